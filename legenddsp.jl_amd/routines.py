"""Detector-level DSP routines: the drop-in surface of the hot path.

`dsp_icpc(data, config, tau, pars_filter)` mirrors reference
src/dsp_icpc.jl:62-230: same required input columns (`waveform`, `baseline`,
`timestamp`, `eventnumber`, `daqenergy`, dsp_icpc.jl:80-84), same 53 output
column names (dsp_icpc.jl:210-229).  The whole chain runs as ONE fused HIP
kernel through `ldsp_icpc_run`; there is no per-filter temporary and no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _abi, _lib
from .config import DSPConfig, lower_icpc, lower_sipm


@dataclass
class ArrayOfRDWaveforms:
    """N traces sharing one time axis (RadiationDetectorSignals.ArrayOfRDWaveforms
    over an ArrayOfSimilarVectors): `signal` is a [N, L] float32 tensor (row = trace),
    the time axis is the range t_first + i*dt in ns."""
    signal: torch.Tensor
    t_first: float = 0.0
    dt: float = 16.0

    def __len__(self):
        return self.signal.shape[0]

    @property
    def nsamples(self):
        return self.signal.shape[1]


class Table(dict):
    """TypedTables.Table stand-in: ordered dict of equally long columns."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __len__(self):
        for v in self.values():
            return len(v)
        return 0

    @property
    def columnnames(self):
        return list(self.keys())


def _as_device_f32(x: torch.Tensor, device) -> torch.Tensor:
    if x.dtype != torch.float32 or not x.is_contiguous() or x.device != device:
        x = x.to(device=device, dtype=torch.float32).contiguous()
    return x


def icpc_run(wf: torch.Tensor, params: _abi.IcpcParams, ctx: _lib.Context = None, out: torch.Tensor = None,
             ext_baseline: torch.Tensor = None, ext_baseline_scale: float = 1.0, main_only: bool = False) -> torch.Tensor:
    """Run the fused kernel on a device-resident [n, L] batch: float32, or uint16 ADC counts (converted by the kernel as
    it loads them — no separate cast pass; any other dtype is cast to float32 here).

    Returns the [n, 48] float32 output table (columns in `_abi.ICPC_COLS` order;
    the 5 integer columns hold int32 bit patterns — see `table_columns`).
    `ext_baseline` (device float32 [n]) x `ext_baseline_scale` replaces the traces' own baseline mean and `main_only`
    skips the CUSP/ZAC stage (`ldsp_icpc_run_opts`; what dsp_icpc_compressed needs for its windowed traces) — per-call
    arguments, the context keeps no state between calls."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "icpc_run needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    in_u16 = wf.dtype == torch.uint16      # raw ADC counts: handed to the kernel as they are (converted while loading)
    wf = wf.contiguous() if in_u16 else _as_device_f32(wf, wf.device)
    nc = len(_abi.ICPC_COLS)
    if out is None:
        out = torch.empty((n, nc), dtype=torch.float32, device=wf.device)
    if out.shape != (n, nc) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != wf.device:
        raise ValueError(f"out must be a contiguous float32 [{n}, {nc}] tensor on the waveforms' device")
    o = _abi.IcpcOut()
    base = out.data_ptr()
    for i, c in enumerate(_abi.ICPC_COLS):
        setattr(o, c, base + 4 * i)
    o.stride = nc
    opts = None
    if ext_baseline is not None or main_only or in_u16:
        if ext_baseline is not None and (ext_baseline.shape != (n,) or ext_baseline.dtype != torch.float32 or ext_baseline.device != wf.device
                                         or not ext_baseline.is_contiguous()):
            raise ValueError("ext_baseline must be a contiguous float32 [n] tensor on the waveforms' device")
        opts = C.byref(_abi.IcpcOpts(None if ext_baseline is None else ext_baseline.data_ptr(), float(ext_baseline_scale), int(bool(main_only)), int(in_u16)))
    ctx.bind_stream()
    _lib.check(_lib.lib().ldsp_icpc_run_opts(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), opts, C.byref(o)))
    return out


def icpc_pz_trap_run(wf: torch.Tensor, params: _abi.IcpcParams, ctx: _lib.Context = None, out: torch.Tensor = None):
    """BASELINE config 2 sub-chain (blmean, e_10410) — `ldsp_icpc_pz_trap_run`. Returns a [2, n] tensor.
    A uint16 tensor (ADC counts) goes to `ldsp_icpc_pz_trap_run_u16` as it is: converted while loading, half the bytes."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    in_u16 = wf.dtype == torch.uint16
    wf = wf.contiguous() if in_u16 else _as_device_f32(wf, wf.device)
    if out is None:
        out = torch.empty((2, n), dtype=torch.float32, device=wf.device)
    if out.shape != (2, n) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != wf.device:
        raise ValueError(f"out must be a contiguous float32 [2, {n}] tensor on the waveforms' device")
    ctx.bind_stream()
    run = _lib.lib().ldsp_icpc_pz_trap_run_u16 if in_u16 else _lib.lib().ldsp_icpc_pz_trap_run
    _lib.check(run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr())))
    return out


def table_columns(tab: torch.Tensor) -> dict:
    """Split the [n, 48] table into named columns (int columns re-viewed as int32)."""
    cols = {}
    ti = tab.view(torch.int32)
    for i, c in enumerate(_abi.ICPC_COLS):
        cols[c] = ti[:, i] if c in _abi.ICPC_I32_COLS else tab[:, i]
    return cols


def dsp_icpc(data: Table, config: DSPConfig, tau: float, pars_filter: dict, f_evaluate_qc=None,
             ctx: _lib.Context = None) -> Table:
    """`dsp_icpc(data::Table, config::DSPConfig, τ, pars_filter::PropDict)` — reference
    src/dsp_icpc.jl:62.  `data.waveform` is an ArrayOfRDWaveforms; returns the 53-column Table."""
    wvfs: ArrayOfRDWaveforms = data["waveform"]
    n = len(wvfs)
    params = lower_icpc(config, tau, pars_filter, wvfs.nsamples, wvfs.t_first, wvfs.dt)
    tab = icpc_run(wvfs.signal, params, ctx)
    c = table_columns(tab)
    dev = tab.device
    res = Table()
    order = ["blmean", "blsigma", "blslope", "bloffset", "tailmean", "tailsigma", "tailslope", "tailoffset"]
    for k in order:
        res[k] = c[k]
    if f_evaluate_qc is None:
        res["qc_label"] = torch.full((n,), -1, dtype=torch.int64, device=dev)  # dsp_icpc.jl:108 (f_evaluate_qc = missing)
    else:   # Int.(get_qc_classifier(wvfs - blmean, f_evaluate_qc))   dsp_icpc.jl:105-108
        from .ml_routines import get_qc_classifier
        res["qc_label"] = torch.as_tensor(get_qc_classifier(wvfs, f_evaluate_qc, config, ctx)).to(torch.int64)
    for k in ["t0", "t10", "t50", "t80", "t90", "t99", "t50_current", "drift_time"]:
        res[k] = c[k]
    res["tail_τ"] = c["tail_tau"]
    for k in ["tail_mean", "tail_sigma", "e_max", "e_min", "e_10410", "e_535", "e_313", "e_10410_inv", "e_313_inv",
              "t0_inv", "e_trap", "e_cusp", "e_zac", "e_trap_max", "e_cusp_max", "e_zac_max",
              "t_trap_max", "t_cusp_max", "t_zac_max", "qdrift", "lq", "a_sg", "a_60", "a_100", "a_raw"]:
        res[k] = c[k]
    res["blfc"] = data["baseline"]          # passthrough columns, dsp_icpc.jl:226
    res["timestamp"] = data["timestamp"]
    res["eventID_fadc"] = data["eventnumber"]
    res["e_fc"] = data["daqenergy"]
    for k in ["inTrace_intersect", "inTrace_n", "n_sat_low", "n_sat_high", "n_sat_low_cons", "n_sat_high_cons"]:
        res[k] = c[k]
    return res


# ---------------------------------------------------------------------------
# dsp_sipm

def sipm_run(wf: torch.Tensor, params: _abi.SipmParams, ctx: _lib.Context = None, out=None, cap: int = None):
    """Run the fused dsp_sipm kernel (`ldsp_sipm_run`).  Returns (scalars [20, n] float32,
    {group: {count [n] int32, x/x_high/x_tot [n, cap] float64, max [n, cap] float32}}), cap = LDSP_MAX_TRIG by default.
    `count` is the TRUE multiplicity: where it exceeds cap the slab holds the first cap triggers only and
    `sipm_resolve_overflow` (called by `dsp_sipm`) runs those traces again with larger slabs.
    `out`: the pair returned by an earlier call on a batch of the same size — its buffers are reused
    without re-initialisation (slab entries beyond `count` then keep their old contents)."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "sipm_run needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    in_u16 = wf.dtype == torch.uint16      # raw ADC counts: handed to the kernel as they are (converted while loading)
    wf = wf.contiguous() if in_u16 else _as_device_f32(wf, wf.device)
    dev = wf.device
    cap = int(cap) if cap else _abi.LDSP_MAX_TRIG
    if out is not None:
        sc, trig = out
        if sc.shape != (len(_abi.SIPM_SCALAR_COLS), n) or sc.device != dev or sc.dtype != torch.float32 or not sc.is_contiguous():
            raise ValueError("out= buffers do not match this batch")
        for g in _abi.SIPM_TRIG_GROUPS:
            t = trig[g]
            t.pop("overflow", None)
            if t["count"].shape != (n,) or t["count"].dtype != torch.int32 or t["count"].device != dev:
                raise ValueError(f"out=: count buffer of group {g} does not match this batch")
            for k in ("x", "x_high", "x_tot", "max"):
                if t[k].shape != (n, t["x"].shape[1]) or t[k].dtype != getattr(torch, _abi.TRIG_DTYPES[k]) or t[k].device != dev or not t[k].is_contiguous():
                    raise ValueError(f"out=: slab {g}.{k} does not match this batch")
    else:
        sc = torch.full((len(_abi.SIPM_SCALAR_COLS), n), float("nan"), dtype=torch.float32, device=dev)
        trig = {}
        for g in _abi.SIPM_TRIG_GROUPS:
            cnt = torch.zeros(n, dtype=torch.int32, device=dev)
            slabs = {k: torch.full((n, cap), float("nan"), dtype=getattr(torch, _abi.TRIG_DTYPES[k]), device=dev) for k in _abi.TRIG_FIELDS}
            trig[g] = dict(count=cnt, **slabs)
    o = _abi.SipmOut()
    for i, c in enumerate(_abi.SIPM_SCALAR_COLS):
        setattr(o, c, sc[i].data_ptr())
    for g in _abi.SIPM_TRIG_GROUPS:
        t = trig[g]
        setattr(o, g, _abi.TrigOut(t["count"].data_ptr(), t["x"].data_ptr(), t["x_high"].data_ptr(), t["x_tot"].data_ptr(), t["max"].data_ptr(),
                                   t["x"].shape[1], 0))
    ctx.bind_stream()
    run = _lib.lib().ldsp_sipm_run_u16 if in_u16 else _lib.lib().ldsp_sipm_run
    _lib.check(run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), C.byref(o)))
    return sc, trig


def sipm_resolve_overflow(wf: torch.Tensor, params: _abi.SipmParams, ctx, trig: dict) -> dict:
    """Second pass of the two-pass trigger fill: traces whose trigger count exceeded the slab capacity run again (the whole
    fused chain, only these traces) with slabs sized from the counts; afterwards `compact_fields` returns every trigger."""
    from .extractors import resolve_overflow
    return resolve_overflow(trig, lambda rows, cap: sipm_run(wf.index_select(0, rows).contiguous(), params, ctx, cap=cap)[1])


def dsp_sipm(data: Table, config: dict, pars_optimization: dict, ctx: _lib.Context = None, _waveform_column="waveform") -> Table:
    """`dsp_sipm(data::Table, config::PropDict, pars_optimization::PropDict)` — reference
    src/dsp_sipm.jl:47-158: 24 scalar columns (4 passthrough) + 12 ragged VectorOfVectors columns,
    names as at dsp_sipm.jl:141-157.  Trigger times are in the time-axis unit (ns)."""
    from .extractors import _compact_group
    wvfs: ArrayOfRDWaveforms = data[_waveform_column]
    params = lower_sipm(config, pars_optimization, wvfs.nsamples, wvfs.t_first, wvfs.dt)
    sc, trig = sipm_run(wvfs.signal, params, ctx)
    trig = sipm_resolve_overflow(wvfs.signal, params, ctx, trig)   # every crossing is returned (src/intersect_maximum.jl:49-56)
    vvs = {g: _compact_group(trig[g], ("x", "x_high", "x_tot", "max"))[0] for g in _abi.SIPM_TRIG_GROUPS}
    s = {c: sc[i] for i, c in enumerate(_abi.SIPM_SCALAR_COLS)}
    res = Table()
    res["blfc"] = data["baseline"]; res["timestamp"] = data["timestamp"]
    res["eventID_fadc"] = data["eventnumber"]; res["e_fc"] = data["daqenergy"]
    for k in ["t_max", "t_min", "t_max_lar", "t_min_lar", "e_max", "e_min", "e_max_lar", "e_min_lar",
              "blmean", "blsigma", "blslope", "bloffset", "wfmean", "wfsigma", "wfslope", "wfoffset"]:
        res[k] = s[k]
    vv = lambda g, f: vvs[g][f]
    res["threshold"] = s["threshold"]; res["threshold_DC"] = s["threshold_DC"]
    res["trig_pos"] = vv("trig", "x"); res["trig_max"] = vv("trig", "max")
    res["trig_pos_DC"] = vv("trig_DC", "x"); res["trig_max_DC"] = vv("trig_DC", "max")
    res["threshold_trap"] = s["threshold_trap"]; res["threshold_DC_trap"] = s["threshold_DC_trap"]
    res["trig_pos_trap"] = vv("trig_trap", "x"); res["trig_pos_high_trap"] = vv("trig_trap", "x_high")
    res["trig_pos_tot_trap"] = vv("trig_trap", "x_tot"); res["trig_max_trap"] = vv("trig_trap", "max")
    res["trig_pos_DC_trap"] = vv("trig_DC_trap", "x"); res["trig_pos_high_DC_trap"] = vv("trig_DC_trap", "x_high")
    res["trig_pos_tot_DC_trap"] = vv("trig_DC_trap", "x_tot"); res["trig_max_DC_trap"] = vv("trig_DC_trap", "max")
    return res


def dsp_sipm_compressed(data: Table, config: dict, pars_optimization: dict, ctx: _lib.Context = None) -> Table:
    """`dsp_sipm_compressed(data, config, pars_optimization)` — reference src/dsp_sipm.jl:207-318: the chain of `dsp_sipm`
    on the `waveform_bit_drop` column.  `decode_data` (LegendDataTypes, :248) is the I/O side's codec: the column is taken
    as a decoded ArrayOfRDWaveforms."""
    return dsp_sipm(data, config, pars_optimization, ctx, _waveform_column="waveform_bit_drop")


# ---------------------------------------------------------------------------
# L3 helper routines composed from the functors (reference src/dsp_routines.jl).  The fused
# kernels do not call these; they are the unfused spelling of the same steps.

def get_t0(wvfs_pz: ArrayOfRDWaveforms, t0_threshold: float, flt_pars=(40.0, 100.0, 2000.0), mintot=1500.0):
    """reference src/dsp_routines.jl:9-25 — returns t0 in us, NaN -> 0."""
    from .filters import TrapezoidalChargeFilter
    from .extractors import Intersect
    flt = TrapezoidalChargeFilter(*flt_pars)(wvfs_pz)
    t0 = Intersect(mintot=mintot)(flt, t0_threshold)["x"] / 1000.0
    return torch.nan_to_num(t0, nan=0.0)


def get_threshold(wvfs: ArrayOfRDWaveforms, threshold, mintot=1000.0):
    """reference src/dsp_routines.jl:33-42 — returns us, NaN -> 0."""
    from .extractors import Intersect
    t = Intersect(mintot=mintot)(wvfs, threshold)["x"] / 1000.0
    return torch.nan_to_num(t, nan=0.0)


def get_qdrift(wvfs: ArrayOfRDWaveforms, t_start_us, dt_range, pol_power=3, sign_est_length=100.0):
    """reference src/dsp_routines.jl:51-64; `dt_range` = (first, last) of the StepRange in ns."""
    from .filters import IntegratorFilter
    from .extractors import SignalEstimator, PolynomialDNI
    integ = IntegratorFilter(1.0)(wvfs)
    est = SignalEstimator(PolynomialDNI(pol_power, sign_est_length))
    t = t_start_us * 1000.0
    e0, e1, e2 = est(integ, t), est(integ, t + dt_range[0]), est(integ, t + dt_range[1])
    return (e2 - e1) - (e1 - e0)


def get_intracePileUp(wvfs: ArrayOfRDWaveforms, sigma_threshold: float, bl_window, mintot=100.0):
    """reference src/dsp_routines.jl:72-82."""
    from .filters import reverse_waveform
    from .extractors import Intersect, signalstats
    thres = signalstats(wvfs, bl_window[0] + wvfs.t_first, bl_window[1])["sigma"] * sigma_threshold
    thres = torch.where(thres == 0, torch.ones_like(thres), thres)
    r = Intersect(mintot=mintot)(reverse_waveform(wvfs), thres)
    last = wvfs.t_first + (wvfs.nsamples - 1) * wvfs.dt
    return dict(intersect=last - r["x"], n=r["multiplicity"])
