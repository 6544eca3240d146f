"""Filter functors — the RadiationDetectorDSP filter protocol on top of the C ABI.

Names, constructor arguments (physical times, here floats in ns) and the protocol
functions mirror the reference: `fltinstance(flt, si)`, `rdfilt_(y, fi, x)` (Julia
`rdfilt!`), `flt_output_length`, `flt_output_time_axis`; `flt(wvfs)` broadcasts over
an ArrayOfRDWaveforms and allocates the output (reference src/derivative.jl:37-55,
src/haar_filter.jl:17-39, src/moving_window_multi.jl:70-129 show the protocol for
in-repo filters; InvCR/Trapezoidal/CUSP/ZAC/SavitzkyGolay/Integrator/Truncate live
in RadiationDetectorDSP and are restated under DESIGN.md assumptions A1-A5).
Every `rdfilt_` is one HIP kernel launch on a [n, L] device batch.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _abi, _lib
from .config import nsamples, sg_npoints, cuspzac_lowered, window_index, _frac, WindowError
from .routines import ArrayOfRDWaveforms

import math


@dataclass(frozen=True)
class SamplingInfo:
    """`smplinfo(wf)`: the shared time axis of a batch."""
    t_first: float
    dt: float
    n: int


def smplinfo(w: ArrayOfRDWaveforms) -> SamplingInfo:
    return SamplingInfo(w.t_first, w.dt, w.nsamples)


def _vp(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _ctx(x: torch.Tensor):
    if not x.is_cuda:
        raise _lib.LdspError(-103, "filter functors need device-resident waveforms (no CPU fallback)")
    ctx = _lib.default_context(x.device.index)
    ctx.bind_stream()
    return ctx


class FilterInstance:
    """What `fltinstance(flt, si)` returns: integer lengths + the input sampling info."""

    def __init__(self, si: SamplingInfo):
        self.si = si

    def flt_input_length(self):
        return self.si.n

    def flt_output_length(self):
        return self.si.n

    def flt_output_time_axis(self):
        """(t_first, dt) of the output range."""
        return self.si.t_first, self.si.dt

    def rdfilt_(self, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class AbstractRadSigFilter:
    def fltinstance(self, si: SamplingInfo) -> FilterInstance:
        raise NotImplementedError

    def __call__(self, wvfs: ArrayOfRDWaveforms) -> ArrayOfRDWaveforms:
        fi = self.fltinstance(smplinfo(wvfs))
        x = wvfs.signal
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        y = torch.empty((x.shape[0], fi.flt_output_length()), dtype=torch.float32, device=x.device)
        fi.rdfilt_(y, x)
        t0, dt = fi.flt_output_time_axis()
        return ArrayOfRDWaveforms(y, t0, dt)


def fltinstance(flt, si):
    return flt.fltinstance(si)


def rdfilt_(y, fi, x):
    """Julia `rdfilt!(y, fi, x)`: writes the caller-allocated y, returns y."""
    assert y.shape[1] == fi.flt_output_length() and x.shape[1] == fi.flt_input_length() and y.shape[0] == x.shape[0]
    return fi.rdfilt_(y, x)


def flt_output_length(fi):
    return fi.flt_output_length()


def flt_input_length(fi):
    return fi.flt_input_length()


def flt_output_time_axis(fi):
    return fi.flt_output_time_axis()


# ---------------------------------------------------------------------------------

class _SimpleInstance(FilterInstance):
    def __init__(self, si, call, n_out=None, t_shift=0, dt_scale=1):
        super().__init__(si)
        self._call, self._n_out, self._t_shift, self._dt_scale = call, n_out, t_shift, dt_scale

    def flt_output_length(self):
        return self.si.n if self._n_out is None else self._n_out

    def flt_output_time_axis(self):
        return self.si.t_first + self._t_shift * self.si.dt, self.si.dt * self._dt_scale

    def rdfilt_(self, y, x):
        ctx = _ctx(x)
        _lib.check(self._call(ctx.handle, _vp(x), x.shape[0], x.shape[1], _vp(y)))
        return y


@dataclass(frozen=True)
class InvCRFilter(AbstractRadSigFilter):
    """`InvCRFilter(tau)` — pole-zero correction y = x + (dt/tau)*cumsum(x)."""
    cr: float

    def fltinstance(self, si):
        c = si.dt / self.cr
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_invcr(h, x, n, L, c, y))


@dataclass(frozen=True)
class IntegratorFilter(AbstractRadSigFilter):
    """`IntegratorFilter(gain)` — gain*cumsum(x)."""
    gain: float = 1.0

    def fltinstance(self, si):
        g = float(self.gain)
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_integrator(h, x, n, L, g, y))


@dataclass(frozen=True)
class TrapezoidalChargeFilter(AbstractRadSigFilter):
    """`TrapezoidalChargeFilter(avgtime, gaptime[, avgtime2])` — valid mode, output stamped
    with the time of the last input sample under the kernel."""
    avgtime: float
    gaptime: float
    avgtime2: float = None

    def fltinstance(self, si):
        a2 = self.avgtime if self.avgtime2 is None else self.avgtime2
        t = _abi.Trap(nsamples(self.avgtime, si.dt), nsamples(self.gaptime, si.dt), nsamples(a2, si.dt))
        if t.navg < 1 or t.navg2 < 1 or t.flen > si.n:
            raise WindowError(f"trapezoid {t} does not fit a trace of {si.n} samples")
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_trap(h, x, n, L, t, y),
                               n_out=si.n - t.flen + 1, t_shift=t.flen - 1)


class _FirInstance(_SimpleInstance):
    def __init__(self, si, h):
        self.h = np.ascontiguousarray(h, dtype=np.float64)
        m = len(self.h)
        if m > si.n:
            raise WindowError("FIR longer than the trace")
        super().__init__(si, lambda hd, x, n, L, y: _lib.lib().ldsp_rdfilt_fir(
            hd, x, n, L, self.h.ctypes.data_as(C.c_void_p), m, y), n_out=si.n - m + 1, t_shift=m - 1)


def _coeffs(fn, cz):
    h = np.empty(cz.length)
    _lib.check(fn(C.byref(cz), h.ctypes.data_as(C.c_void_p)))
    return h


@dataclass(frozen=True)
class CUSPChargeFilter(AbstractRadSigFilter):
    """`CUSPChargeFilter(sigma, toplen, tau, length, beta)` (reference call site src/dsp_icpc.jl:167)."""
    sigma: float
    toplen: float
    tau: float
    length: float
    beta: float

    def fltinstance(self, si):
        cz = cuspzac_lowered(self.sigma, self.toplen, self.tau, self.length, self.beta, si.dt)
        return _FirInstance(si, _coeffs(_lib.lib().ldsp_cusp_coeffs, cz))


@dataclass(frozen=True)
class ZACChargeFilter(AbstractRadSigFilter):
    """`ZACChargeFilter(sigma, toplen, tau, length, beta)` (reference call site src/dsp_icpc.jl:174)."""
    sigma: float
    toplen: float
    tau: float
    length: float
    beta: float

    def fltinstance(self, si):
        cz = cuspzac_lowered(self.sigma, self.toplen, self.tau, self.length, self.beta, si.dt)
        return _FirInstance(si, _coeffs(_lib.lib().ldsp_zac_coeffs, cz))


@dataclass(frozen=True)
class SavitzkyGolayFilter(AbstractRadSigFilter):
    """`SavitzkyGolayFilter(length, degree, derivative)` (reference call sites src/dsp_icpc.jl:181-185)."""
    length: float
    degree: int
    derivative: int = 0

    def fltinstance(self, si):
        n = sg_npoints(self.length, si.dt)
        h = np.empty(n)
        _lib.check(_lib.lib().ldsp_sg_coeffs(n, int(self.degree), int(self.derivative), h.ctypes.data_as(C.c_void_p)))
        return _FirInstance(si, h)


@dataclass(frozen=True)
class DerivativeFilter(AbstractRadSigFilter):
    """`DerivativeFilter(gain)` — reference src/derivative.jl:26-55."""
    gain: float = 1.0

    def fltinstance(self, si):
        g = float(self.gain)
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_derivative(h, x, n, L, g, y))


@dataclass(frozen=True)
class HaarAveragingFilter(AbstractRadSigFilter):
    """`HaarAveragingFilter(down_sampling_rate)` — reference src/haar_filter.jl:3-39."""
    down_sampling_rate: int

    def fltinstance(self, si):
        ds = int(self.down_sampling_rate)
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_haar(h, x, n, L, ds, y),
                               n_out=-(-si.n // ds), dt_scale=ds)


@dataclass(frozen=True)
class MovingWindowFilter(AbstractRadSigFilter):
    """`MovingWindowFilter(length)` — reference src/moving_window_multi.jl:58-116."""
    length: float

    def fltinstance(self, si):
        l = nsamples(self.length, si.dt)
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_moving_window(h, x, n, L, l, y))


@dataclass(frozen=True)
class MovingWindowMultiFilter(AbstractRadSigFilter):
    """`MovingWindowMultiFilter(length)` — reference src/moving_window_multi.jl:29-32,118-129."""
    length: float

    def fltinstance(self, si):
        l = nsamples(self.length, si.dt)
        return _SimpleInstance(si, lambda h, x, n, L, y: _lib.lib().ldsp_rdfilt_moving_window_multi(h, x, n, L, l, y))


def _affine(x, L, frm, until, scale, shift, per_trace, rev, n_out):
    def call(h, xp, n, Lx, y):
        return _lib.lib().ldsp_rdfilt_affine(h, xp, n, Lx, frm, until, float(scale), float(shift),
                                             _vp(per_trace) if per_trace is not None else None, int(rev), y)
    return call


@dataclass(frozen=True)
class TruncateFilter(AbstractRadSigFilter):
    """`TruncateFilter(a..b)` — the samples whose time lies in the closed interval."""
    left: float
    right: float

    def fltinstance(self, si):
        fa = (_frac(self.left) - _frac(si.t_first)) / _frac(si.dt)
        fb = (_frac(self.right) - _frac(si.t_first)) / _frac(si.dt)
        a, b = max(0, math.ceil(fa)), min(si.n - 1, math.floor(fb))
        if not (0 <= a <= b):
            raise WindowError("TruncateFilter interval outside the trace")
        return _SimpleInstance(si, _affine(None, si.n, a, b, 1.0, 0.0, None, 0, b - a + 1), n_out=b - a + 1, t_shift=a)


class TimeAxisFilter(AbstractRadSigFilter):
    """`TimeAxisFilter(period, offset = 0)` — reference src/timeaxis.jl:31-60: the sampling step of the time axis becomes `period`,
    its first point moves by `offset`; the samples are unchanged (`rdfilt!` copies them, :60 — here the output shares the input's
    storage unless the caller supplies y).  Only range time axes exist in this package (the `ArgumentError` of :57 has no trigger)."""

    def __init__(self, period: float, offset: float = 0.0):
        self.period, self.offset = float(period), float(offset)

    def fltinstance(self, si: SamplingInfo) -> FilterInstance:
        flt = self

        class _Inst(FilterInstance):
            def flt_output_time_axis(self):
                return self.si.t_first + flt.offset, flt.period

            def rdfilt_(self, y, x):
                if y.data_ptr() != x.data_ptr():
                    y.copy_(x)
                return y
        return _Inst(si)

    def __call__(self, wvfs: ArrayOfRDWaveforms) -> ArrayOfRDWaveforms:
        t0, dt = self.fltinstance(smplinfo(wvfs)).flt_output_time_axis()
        return ArrayOfRDWaveforms(wvfs.signal, t0, dt)


def shift_waveform(wvfs: ArrayOfRDWaveforms, c) -> ArrayOfRDWaveforms:
    """`shift_waveform.(wvfs, c)` — c a scalar or one value per trace (src/dsp_icpc.jl:105)."""
    per = c if isinstance(c, torch.Tensor) else None
    sh = 0.0 if per is not None else float(c)
    if per is not None:
        per = per.to(device=wvfs.signal.device, dtype=torch.float32).contiguous()
    si = smplinfo(wvfs)
    fi = _SimpleInstance(si, _affine(None, si.n, 0, si.n - 1, 1.0, sh, per, 0, si.n))
    y = torch.empty_like(wvfs.signal, dtype=torch.float32)
    fi.rdfilt_(y, wvfs.signal.to(torch.float32).contiguous())
    return ArrayOfRDWaveforms(y, wvfs.t_first, wvfs.dt)


def multiply_waveform(wvfs: ArrayOfRDWaveforms, c: float) -> ArrayOfRDWaveforms:
    """`multiply_waveform.(wvfs, c)` (src/dsp_icpc.jl:199)."""
    si = smplinfo(wvfs)
    fi = _SimpleInstance(si, _affine(None, si.n, 0, si.n - 1, float(c), 0.0, None, 0, si.n))
    y = torch.empty_like(wvfs.signal, dtype=torch.float32)
    fi.rdfilt_(y, wvfs.signal.to(torch.float32).contiguous())
    return ArrayOfRDWaveforms(y, wvfs.t_first, wvfs.dt)


def reverse_waveform(wvfs: ArrayOfRDWaveforms) -> ArrayOfRDWaveforms:
    """`reverse_waveform.(wvfs)`: reversed signal, unchanged time axis (src/dsp_routines.jl:79)."""
    si = smplinfo(wvfs)
    fi = _SimpleInstance(si, _affine(None, si.n, 0, si.n - 1, 1.0, 0.0, None, 1, si.n))
    y = torch.empty_like(wvfs.signal, dtype=torch.float32)
    fi.rdfilt_(y, wvfs.signal.to(torch.float32).contiguous())
    return ArrayOfRDWaveforms(y, wvfs.t_first, wvfs.dt)
