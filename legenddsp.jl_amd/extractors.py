"""Feature extractors — callable structs / functions returning per-trace columns.

The reference's extractors take one waveform and return a NamedTuple; broadcast over
an ArrayOfRDWaveforms yields a struct-of-arrays whose fields are columns
(`inters.x`, `estats.max`, reference src/dsp_routines.jl:21, src/dsp_sipm.jl:149).  Here a
call on a batch returns a dict of [n] device tensors directly; ragged fields come back as
`VectorOfVectors` (offsets + values).  Times are floats in ns.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _abi, _lib
from .config import nsamples, window_index, WindowError
from .routines import ArrayOfRDWaveforms


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _prep(w: ArrayOfRDWaveforms):
    x = w.signal
    if not x.is_cuda:
        raise _lib.LdspError(-103, "extractors need device-resident waveforms (no CPU fallback)")
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    ctx = _lib.default_context(x.device.index)
    ctx.bind_stream()
    return ctx, x


def _f(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


def _i(n, dev):
    return torch.empty(n, dtype=torch.int32, device=dev)


def _window(w, start, stop):
    a, b = window_index(start, w.t_first, w.dt), window_index(stop, w.t_first, w.dt)
    if not (0 <= a <= b <= w.nsamples - 1):
        raise WindowError(f"window [{a},{b}] outside a trace of {w.nsamples} samples")  # the reference's @assert
    return a, b


def _per_trace(v, n, dev):
    """a per-trace argument (threshold, time) as a device float32 [n]: a tensor of n elements, or a scalar for every trace"""
    if isinstance(v, torch.Tensor):
        if v.numel() == 1:
            return torch.full((n,), float(v), dtype=torch.float32, device=dev)
        if v.shape != (n,):      # the kernel reads element i for trace i: anything else would read out of bounds
            raise ValueError(f"per-trace argument has shape {tuple(v.shape)}, expected ({n},) or a scalar")
        return v.to(device=dev, dtype=torch.float32).contiguous()
    return torch.full((n,), float(v), dtype=torch.float32, device=dev)


@dataclass
class VectorOfVectors:
    """ArraysOfArrays.VectorOfVectors: `values[offsets[i]:offsets[i+1]]` is element i."""
    offsets: torch.Tensor
    values: torch.Tensor

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        return self.values[int(self.offsets[i]):int(self.offsets[i + 1])]


class TriggerOverflow(_lib.LdspError):
    """A trace produced more triggers than its slab holds and the caller gave no way to run it again."""


def compact_fields(t: dict, fields=("x", "x_high", "x_tot", "max")):
    """Compact the slabs of one trigger group (`count` [n], one [n, cap] slab per field, optionally an `overflow`
    record = the same for the traces whose count exceeded cap, re-run with larger slabs by `resolve_overflow`) into
    `(values [sum(count), len(fields)] float64, count [n] int64)` (positions are Float64 slabs, `max` float32 ones: one
    float64 block holds both exactly): EVERY trigger of every trace, in trace order — what the
    reference's `VectorOfVectors` columns hold (src/intersect_maximum.jl:49-56 pushes every crossing).  Work is
    proportional to the number of triggers, not to n x cap.  Raises TriggerOverflow if a count exceeds its slab and no
    overflow record covers it."""
    count = t["count"].to(torch.int64)
    n, dev = len(count), count.device
    cap = t[fields[0]].shape[1]
    ovf = t.get("overflow")
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(count, 0)
    total = int(offsets[-1])
    row = torch.repeat_interleave(torch.arange(n, device=dev), count, output_size=total)
    pos = torch.arange(total, device=dev) - offsets[row]
    over = pos >= cap
    if ovf is None:
        if total and bool(over.any()):
            raise TriggerOverflow(-104, f"a trace has more than {cap} triggers: call resolve_overflow() first")
        vals = torch.stack([t[f][row, pos].to(torch.float64) for f in fields], dim=1) if total else torch.empty((0, len(fields)), dtype=torch.float64, device=dev)
        return vals, count
    slot = torch.full((n,), -1, dtype=torch.int64, device=dev)
    slot[ovf["rows"]] = torch.arange(len(ovf["rows"]), device=dev)
    use2 = slot[row] >= 0                      # every element of a re-run trace comes from the larger slabs
    if bool((over & ~use2).any()) or bool((ovf["count"].to(torch.int64) != count[ovf["rows"]]).any()):
        raise TriggerOverflow(-104, "overflow record does not cover every overflowing trace")
    cols = []
    for f in fields:
        a = t[f][row, pos.clamp(max=cap - 1)]
        b = ovf[f][slot[row].clamp(min=0), pos.clamp(max=ovf[f].shape[1] - 1)]
        cols.append(torch.where(use2, b, a).to(torch.float64))
    return torch.stack(cols, dim=1), count


def _compact_group(t: dict, fields):
    """-> ({field: VectorOfVectors}, count) of one trigger group (shared offsets)."""
    vals, count = compact_fields(t, fields)
    offsets = torch.zeros(len(count) + 1, dtype=torch.int64, device=count.device)
    offsets[1:] = torch.cumsum(count, 0)
    return {f: VectorOfVectors(offsets, vals[:, i].contiguous().to(getattr(torch, _abi.TRIG_DTYPES.get(f, "float64"))))
            for i, f in enumerate(fields)}, count


def resolve_overflow(groups: dict, rerun):
    """Two-pass count-then-fill.  `groups`: {name: trigger-group dict}.  If any trace of any group counted more triggers
    than its slab holds, `rerun(rows, cap)` must run exactly those traces again with slabs of capacity `cap` and return
    the same structure; its result is attached as the `overflow` record `compact_fields` reads.  One host read."""
    names = list(groups)
    cap = {g: groups[g]["x"].shape[1] for g in names}
    mx = torch.stack([groups[g]["count"].max() if len(groups[g]["count"]) else torch.zeros((), dtype=torch.int32, device=groups[g]["count"].device)
                      for g in names]).tolist()
    if all(m <= cap[g] for m, g in zip(mx, names)):
        return groups
    over = None
    for g in names:
        o = groups[g]["count"] > cap[g]
        over = o if over is None else (over | o)
    rows = torch.nonzero(over)[:, 0]
    cap2 = 1 << (int(max(mx)) - 1).bit_length()
    again = rerun(rows, cap2)
    for g in names:
        groups[g]["overflow"] = dict(rows=rows, **{k: v for k, v in again[g].items() if k != "overflow"})
    return groups


def signalstats(w: ArrayOfRDWaveforms, start, stop):
    """`signalstats(wf, start, stop)` -> mean, sigma, slope, offset."""
    ctx, x = _prep(w)
    a, b = _window(w, start, stop)
    n, dev = x.shape[0], x.device
    o = [_f(n, dev) for _ in range(4)]
    _lib.check(_lib.lib().ldsp_signalstats(ctx.handle, _vp(x), n, x.shape[1], a, b, w.t_first, w.dt, *[_vp(t) for t in o]))
    return dict(mean=o[0], sigma=o[1], slope=o[2], offset=o[3])


def tailstats(w: ArrayOfRDWaveforms, start, stop):
    """`tailstats(wf, start, stop)` -> mean, sigma, τ (reference src/tailstats.jl:13-72)."""
    ctx, x = _prep(w)
    a, b = _window(w, start, stop)
    n, dev = x.shape[0], x.device
    o = [_f(n, dev) for _ in range(3)]
    _lib.check(_lib.lib().ldsp_tailstats(ctx.handle, _vp(x), n, x.shape[1], a, b, w.t_first, w.dt, *[_vp(t) for t in o]))
    return {"mean": o[0], "sigma": o[1], "τ": o[2]}


def extremestats(w: ArrayOfRDWaveforms, start=None, stop=None):
    """`extremestats(wf[, start, stop])` -> min, max, tmin, tmax (reference src/extremestats.jl:14-40)."""
    ctx, x = _prep(w)
    if start is None:
        a, b = 0, w.nsamples - 1
    else:
        a, b = _window(w, start, stop)
    n, dev = x.shape[0], x.device
    o = [_f(n, dev) for _ in range(4)]
    _lib.check(_lib.lib().ldsp_extremestats(ctx.handle, _vp(x), n, x.shape[1], a, b, w.t_first, w.dt, *[_vp(t) for t in o]))
    return dict(min=o[0], max=o[1], tmin=o[2], tmax=o[3])


def thresholdstats(w: ArrayOfRDWaveforms, min=-float("inf"), max=float("inf")):
    """reference src/thresholdstats.jl:14-41"""
    ctx, x = _prep(w)
    o = _f(x.shape[0], x.device)
    _lib.check(_lib.lib().ldsp_thresholdstats(ctx.handle, _vp(x), x.shape[0], x.shape[1], float(min), float(max), _vp(o)))
    return o


def thresholdstats_mad(w: ArrayOfRDWaveforms, min=-float("inf"), max=float("inf")):
    """reference src/thresholdstats.jl:56-71"""
    ctx, x = _prep(w)
    o = _f(x.shape[0], x.device)
    _lib.check(_lib.lib().ldsp_thresholdstats_mad(ctx.handle, _vp(x), x.shape[0], x.shape[1], float(min), float(max), _vp(o)))
    return o


def saturation(w: ArrayOfRDWaveforms, low, high, start=None, stop=None):
    """`saturation(wf, low, high)` / `saturation(wf, start, stop, low, high)` (reference src/saturation.jl:12-65)."""
    ctx, x = _prep(w)
    a, b = (0, w.nsamples - 1) if start is None else _window(w, start, stop)
    n, dev = x.shape[0], x.device
    o = [_i(n, dev) for _ in range(4)]
    _lib.check(_lib.lib().ldsp_saturation(ctx.handle, _vp(x), n, x.shape[1], a, b, float(low), float(high), *[_vp(t) for t in o]))
    return dict(low=o[0], high=o[1], max_cons_low=o[2], max_cons_high=o[3])


def get_wvf_maximum(w: ArrayOfRDWaveforms, start, stop):
    """reference src/interpolation.jl:21-46"""
    ctx, x = _prep(w)
    a, b = _window(w, start, stop)
    o = _f(x.shape[0], x.device)
    _lib.check(_lib.lib().ldsp_get_wvf_maximum(ctx.handle, _vp(x), x.shape[0], x.shape[1], a, b, _vp(o)))
    return o


@dataclass(frozen=True)
class Intersect:
    """`Intersect(mintot = ...)` -> (x, multiplicity); x = NaN where no crossing."""
    mintot: float = 4.0

    def __call__(self, w: ArrayOfRDWaveforms, threshold):
        ctx, x = _prep(w)
        n, dev = x.shape[0], x.device
        thr = _per_trace(threshold, n, dev)
        min_n = max(1, nsamples(self.mintot, w.dt))
        xo, mult = _f(n, dev), _i(n, dev)
        _lib.check(_lib.lib().ldsp_intersect(ctx.handle, _vp(x), n, x.shape[1], w.t_first, w.dt, _vp(thr), min_n, _vp(xo), _vp(mult)))
        return dict(x=xo, multiplicity=mult)


@dataclass(frozen=True)
class IntersectMaximum:
    """`IntersectMaximum(mintot, maxtot)` (reference src/intersect_maximum.jl:11-119)."""
    mintot: float = 4.0
    maxtot: float = 100.0

    def __call__(self, w: ArrayOfRDWaveforms, threshold):
        ctx, x = _prep(w)
        n, dev = x.shape[0], x.device
        if x.shape[1] == 0:   # empty waveforms: empty vectors, multiplicity 0 (reference test/test_intersect_maximum.jl:81-90)
            empty = lambda f: VectorOfVectors(torch.zeros(n + 1, dtype=torch.int64, device=dev),
                                              torch.empty(0, dtype=getattr(torch, _abi.TRIG_DTYPES[f]), device=dev))
            return dict(x=empty("x"), x_high=empty("x_high"), x_tot=empty("x_tot"), max=empty("max"),
                        multiplicity=torch.zeros(n, dtype=torch.int32, device=dev))
        thr = _per_trace(threshold, n, dev)
        min_n, max_n = max(1, nsamples(self.mintot, w.dt)), max(1, nsamples(self.maxtot, w.dt))
        fields = ("x", "x_high", "x_tot", "max")

        def run(xs, ths, cap):
            m = xs.shape[0]
            count = _i(m, dev)
            slabs = {k: torch.full((m, cap), float("nan"), dtype=getattr(torch, _abi.TRIG_DTYPES[k]), device=dev) for k in fields}
            o = _abi.TrigOut(count.data_ptr(), *[slabs[k].data_ptr() for k in fields], cap, 0)
            _lib.check(_lib.lib().ldsp_intersect_maximum(ctx.handle, _vp(xs), m, xs.shape[1], w.t_first, w.dt, _vp(ths), min_n, max_n, C.byref(o)))
            return dict(count=count, **slabs)

        # every crossing is returned (src/intersect_maximum.jl:49-56): traces with more triggers than the default slab run again
        grp = resolve_overflow({"t": run(x, thr, _abi.LDSP_MAX_TRIG)},
                               lambda rows, cap: {"t": run(x.index_select(0, rows).contiguous(), thr.index_select(0, rows).contiguous(), cap)})["t"]
        res, count = _compact_group(grp, fields)
        res["multiplicity"] = grp["count"]
        return res


@dataclass(frozen=True)
class MultiIntersect:
    """`MultiIntersect(threshold_ratios, mintot, n, d, sampling_rate)` (reference src/multi_intersect.jl:11-125)."""
    threshold_ratios: tuple = tuple(np.arange(0.01, 0.905, 0.01))
    mintot: float = 4.0
    n: int = 1
    d: int = 1
    sampling_rate: int = 1

    def __call__(self, w: ArrayOfRDWaveforms):
        ctx, x = _prep(w)
        nt, dev = x.shape[0], x.device
        r = np.ascontiguousarray(self.threshold_ratios, dtype=np.float64)
        K = len(r)
        min_n = max(1, nsamples(self.mintot, w.dt))
        out = torch.empty((nt, K), dtype=torch.float32, device=dev)
        status = _i(nt, dev)
        _lib.check(_lib.lib().ldsp_multi_intersect(ctx.handle, _vp(x), nt, x.shape[1], w.t_first, w.dt, r.ctypes.data_as(C.c_void_p),
                                                   K, min_n, int(self.n), int(self.d), int(self.sampling_rate), _vp(out), _vp(status)))
        if bool((status != 0).any()):
            raise WindowError("cannot interpolate intersect on left boundary")  # reference src/multi_intersect.jl:77-78
        return out


@dataclass(frozen=True)
class PolynomialDNI:
    degree: int
    length: float


@dataclass(frozen=True)
class SignalEstimator:
    """`SignalEstimator(PolynomialDNI(degree, length))(wf, t)` (assumption A3)."""
    method: PolynomialDNI

    def __call__(self, w: ArrayOfRDWaveforms, t):
        ctx, x = _prep(w)
        n, dev = x.shape[0], x.device
        tt = _per_trace(t, n, dev)
        est = _abi.Dni(nsamples(self.method.length, w.dt), int(self.method.degree))
        o = _f(n, dev)
        _lib.check(_lib.lib().ldsp_signal_estimator(ctx.handle, _vp(x), n, x.shape[1], w.t_first, w.dt, _vp(tt), est, _vp(o)))
        return o
