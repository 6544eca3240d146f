"""Filter-optimisation grid scans (reference src/dsp_filter_optimization.jl) — SURVEY 8(f) row 1, trapezoid part.

`dsp_trap_rt_optimization(wvfs, config, tau; ft)` (:102-133) and `dsp_trap_ft_optimization(wvfs, config, tau, rt)`
(:241-274): baseline subtraction, pole-zero deconvolution, then for every grid value the `SignalEstimator` of the
trapezoid-filtered trace at a pick-off.  Here one kernel launch (`ldsp_trap_grid_run`) reads each trace once and
returns the dense `[grid, n]` float32 matrix (the reference: Float64 / Float32 matrices of the same shape).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi, _lib
from .config import DSPConfig, WindowError, cuspzac_lowered, get_fltpars, nsamples, sg_npoints, trap_samples, window_index, UNIT_PER_US
from .routines import ArrayOfRDWaveforms, Table, _as_device_f32


def lower_trap_grid(config: DSPConfig, tau: float, L: int, t_first: float, dt: float, pick_mode: int, pick_time: float = 0.0):
    kw = config.kwargs_pars
    p = _abi.TrapGridParams()
    p.L, p.t_first, p.dt = int(L), float(t_first), float(dt)
    p.bl_from = window_index(config.bl_window.left, t_first, dt)
    p.bl_until = window_index(config.bl_window.right, t_first, dt)
    if not (0 <= p.bl_from <= p.bl_until <= L - 1):
        raise WindowError(f"bl_window [{p.bl_from},{p.bl_until}] outside a trace of {L} samples")
    p.pz_c = float(dt) / float(tau)
    p.sig_est = _abi.Dni(nsamples(kw.sig_interpolation_length, dt), int(kw.sig_interpolation_order))
    p.pick_mode, p.tx_mintot, p.pick_time = int(pick_mode), max(1, nsamples(kw.tx_mintot, dt)), float(pick_time)
    return p


def trap_grid_run(wf: torch.Tensor, params: _abi.TrapGridParams, traps, offsets=None, ctx: _lib.Context = None) -> torch.Tensor:
    """`ldsp_trap_grid_run`: [G, n] float32 device tensor."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "trap_grid_run needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    wf = _as_device_f32(wf, wf.device)
    G = len(traps)
    tr = (_abi.Trap * G)(*traps)
    offs = None
    if offsets is not None:
        offs = np.ascontiguousarray(offsets, dtype=np.float64)
        assert len(offs) == G
    out = torch.empty((G, n), dtype=torch.float32, device=wf.device)
    ctx.bind_stream()
    _lib.check(_lib.lib().ldsp_trap_grid_run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), G, C.cast(tr, C.c_void_p),
                                             offs.ctypes.data_as(C.c_void_p) if offs is not None else None, C.c_void_p(out.data_ptr())))
    return out


def dsp_trap_rt_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, ft: float = 2000.0, ctx=None) -> torch.Tensor:
    """ENC grid over the trapezoid rise times `config.e_grid_rt_trap` at flat-top `ft` (ns); pick-off at
    `config.enc_pickoff_trap` (reference :102-133).  Returns `[len(grid), n]`."""
    grid = list(config.e_grid_rt_trap)
    p = lower_trap_grid(config, tau, wvfs.nsamples, wvfs.t_first, wvfs.dt, 0, config.enc_pickoff_trap)
    traps = [trap_samples(rt, ft, wvfs.dt) for rt in grid]
    return trap_grid_run(wvfs.signal, p, traps, None, ctx)


def dsp_trap_ft_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, rt: float, ctx=None) -> torch.Tensor:
    """Energy grid over the flat-top times `config.e_grid_ft_trap` at rise time `rt` (ns); pick-off at
    t50 + rt + ft/2 (reference :241-274).  Returns `[len(grid), n]`."""
    grid = list(config.e_grid_ft_trap)
    p = lower_trap_grid(config, tau, wvfs.nsamples, wvfs.t_first, wvfs.dt, 1)
    traps = [trap_samples(rt, ft, wvfs.dt) for ft in grid]
    offsets = [float(rt) + float(ft) / 2 for ft in grid]
    return trap_grid_run(wvfs.signal, p, traps, offsets, ctx)


def fir_grid_run(wf: torch.Tensor, params: _abi.TrapGridParams, taps: np.ndarray, offsets=None, ctx: _lib.Context = None) -> torch.Tensor:
    """`ldsp_fir_grid_run`: taps [G, Lf] float64 (host, FIR order) -> [G, n] float32 device tensor."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "fir_grid_run needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    wf = _as_device_f32(wf, wf.device)
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    G, Lf = taps.shape
    offs = None
    if offsets is not None:
        offs = np.ascontiguousarray(offsets, dtype=np.float64)
        assert len(offs) == G
    out = torch.empty((G, n), dtype=torch.float32, device=wf.device)
    ctx.bind_stream()
    _lib.check(_lib.lib().ldsp_fir_grid_run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), G, Lf, taps.ctypes.data_as(C.c_void_p),
                                            offs.ctypes.data_as(C.c_void_p) if offs is not None else None, C.c_void_p(out.data_ptr())))
    return out


_TAU_OFF = 10000000.0 * 1000.0   # 1e7 us in ns: "switch off the CR filter" (dsp_filter_optimization.jl:153)


def cuspzac_grid_taps(kind: str, pairs, length: float, dt: float) -> np.ndarray:
    """Coefficients of CUSPChargeFilter / ZACChargeFilter(rt, ft, tau_off, length, length/dt) for every (rt, ft) pair."""
    fn = _lib.lib().ldsp_cusp_coeffs if kind == "cusp" else _lib.lib().ldsp_zac_coeffs
    rows = []
    for rt, ft in pairs:
        cz = cuspzac_lowered(rt, ft, _TAU_OFF, length, float(length) / float(dt), dt)
        h = np.empty(cz.length, dtype=np.float64)
        _lib.check(fn(C.byref(cz), h.ctypes.data_as(C.c_void_p)))
        rows.append(h)
    return np.stack(rows)


def _cz_rt(kind, wvfs, config, tau, ft, ctx):
    grid = list(getattr(config, f"e_grid_rt_{kind}"))
    length = getattr(config, f"flt_length_{kind}")
    p = lower_trap_grid(config, tau, wvfs.nsamples, wvfs.t_first, wvfs.dt, 0, getattr(config, f"enc_pickoff_{kind}"))
    return fir_grid_run(wvfs.signal, p, cuspzac_grid_taps(kind, [(rt, ft) for rt in grid], length, wvfs.dt), None, ctx)


def _cz_ft(kind, wvfs, config, tau, rt, ctx):
    grid = list(getattr(config, f"e_grid_ft_{kind}"))
    length = getattr(config, f"flt_length_{kind}")
    p = lower_trap_grid(config, tau, wvfs.nsamples, wvfs.t_first, wvfs.dt, 1)
    return fir_grid_run(wvfs.signal, p, cuspzac_grid_taps(kind, [(rt, ft) for ft in grid], length, wvfs.dt), [float(length) / 2] * len(grid), ctx)


def dsp_cusp_rt_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, ft: float = 2000.0, ctx=None) -> torch.Tensor:
    """ENC grid over `config.e_grid_rt_cusp` at flat top `ft`, pick-off `config.enc_pickoff_cusp` (reference :145-181)."""
    return _cz_rt("cusp", wvfs, config, tau, ft, ctx)


def dsp_zac_rt_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, ft: float = 2000.0, ctx=None) -> torch.Tensor:
    """The same with ZACChargeFilter (reference :193-229)."""
    return _cz_rt("zac", wvfs, config, tau, ft, ctx)


def dsp_cusp_ft_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, rt: float, ctx=None) -> torch.Tensor:
    """Energy grid over `config.e_grid_ft_cusp` at rise time `rt`, pick-off t50 + flt_length_cusp/2 (reference :286-324)."""
    return _cz_ft("cusp", wvfs, config, tau, rt, ctx)


def dsp_zac_ft_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, rt: float, ctx=None) -> torch.Tensor:
    """The same with ZACChargeFilter (reference :336-374)."""
    return _cz_ft("zac", wvfs, config, tau, rt, ctx)


def dsp_sg_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, pars_filter: dict, f_evaluate_qc=None, ctx=None) -> Table:
    """`dsp_sg_optimization(wvfs, config, tau, pars_filter)` (reference :393-441): A/E over the Savitzky-Golay window
    lengths `config.a_grid_wl_sg`.  Columns: aoe [W, n], energy, blmean, blslope, t50 (us), qc_label (-1: no classifier)."""
    x = wvfs.signal
    if not x.is_cuda:
        raise _lib.LdspError(-103, "dsp_sg_optimization needs device-resident waveforms (no CPU fallback)")
    ctx = ctx or _lib.default_context(x.device.index)
    x = _as_device_f32(x, x.device)
    n, L = x.shape
    dt, t0 = wvfs.dt, wvfs.t_first
    rt, ft = get_fltpars(pars_filter, "trap", config)
    p = lower_trap_grid(config, tau, L, t0, dt, 1)
    grid = list(config.a_grid_wl_sg)
    W = len(grid)
    npts = np.array([sg_npoints(wl, dt) for wl in grid], dtype=np.int32)
    # current window on each filter's output axis (trailing alignment: its first time is t0 + (npts-1) dt)
    frm = np.array([window_index(config.current_window.left, t0 + (m - 1) * dt, dt) for m in npts], dtype=np.int32)
    until = np.array([window_index(config.current_window.right, t0 + (m - 1) * dt, dt) for m in npts], dtype=np.int32)
    for a, b, m in zip(frm, until, npts):
        if not (0 <= a <= b <= L - m):
            raise WindowError(f"current_window [{a},{b}] outside the output of a {m}-point Savitzky-Golay filter")
    dev = x.device
    amax = torch.empty((W, n), dtype=torch.float32, device=dev)
    energy, t50, blmean, blslope = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    trap = trap_samples(rt, ft, dt)
    ctx.bind_stream()
    vp = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().ldsp_sg_grid_run(ctx.handle, vp(x), n, C.byref(p), C.byref(trap), float(rt) + float(ft) / 2, UNIT_PER_US, W,
                                           npts.ctypes.data_as(C.c_void_p), int(config.sg_flt_degree), frm.ctypes.data_as(C.c_void_p),
                                           until.ctypes.data_as(C.c_void_p), vp(amax), vp(energy), vp(t50), vp(blmean), vp(blslope)))
    res = Table()
    res["aoe"] = amax / energy[None, :]
    res["energy"], res["blmean"], res["blslope"], res["t50"] = energy, blmean, blslope, t50
    if f_evaluate_qc is None:
        res["qc_label"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
    else:   # get_qc_classifier on the baseline-subtracted traces (reference :408-411)
        from .ml_routines import get_qc_classifier
        res["qc_label"] = torch.as_tensor(get_qc_classifier(wvfs, f_evaluate_qc, config, ctx)).to(torch.int32)
    return res


def dsp_qc_flt_optimization(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, f_evaluate_qc=None, ctx=None, _compressed=False,
                            _trap=None) -> Table:
    """`dsp_qc_flt_optimization(wvfs, config, tau, missing)` (reference :9-63): energy with the default trapezoid,
    baseline mean / slope, t50 (us), qc_label (-1 without a classifier).  One launch of `ldsp_sg_grid_run` with an empty grid."""
    x = wvfs.signal
    if not x.is_cuda:
        raise _lib.LdspError(-103, "dsp_qc_flt_optimization needs device-resident waveforms (no CPU fallback)")
    ctx = ctx or _lib.default_context(x.device.index)
    x = _as_device_f32(x, x.device)
    n, L = x.shape
    rt, ft = _trap if _trap is not None else get_fltpars({}, "trap", config)      # config.default_flt_param.trap
    p = lower_trap_grid(config, tau, L, wvfs.t_first, wvfs.dt, 1)
    dev = x.device
    energy, t50, blmean, blslope = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    trap = trap_samples(rt, ft, wvfs.dt)
    ctx.bind_stream()
    vp = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().ldsp_sg_grid_run(ctx.handle, vp(x), n, C.byref(p), C.byref(trap), float(rt) + float(ft) / 2, UNIT_PER_US, 0,
                                           None, int(config.sg_flt_degree), None, None, None, vp(energy), vp(t50), vp(blmean), vp(blslope)))
    res = Table()
    res["energy"], res["blmean"], res["blslope"], res["t50"] = energy, blmean, blslope, t50
    if f_evaluate_qc is None:
        res["qc_label"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
    else:   # get_qc_classifier on the pole-zero corrected traces (reference :31-49)
        from .filters import InvCRFilter, shift_waveform
        from .ml_routines import get_qc_classifier, get_qc_classifier_compressed
        w_pz = InvCRFilter(float(tau))(shift_waveform(ArrayOfRDWaveforms(x, wvfs.t_first, wvfs.dt), -blmean))
        get_qc = get_qc_classifier_compressed if _compressed else get_qc_classifier
        res["qc_label"] = torch.as_tensor(get_qc(w_pz, f_evaluate_qc, None, ctx)).to(torch.int32)
    return res


def dsp_qc_flt_optimization_compressed(wvfs: ArrayOfRDWaveforms, config: DSPConfig, tau: float, f_evaluate_qc=None, ctx=None) -> Table:
    """`dsp_qc_flt_optimization_compressed` (reference :23-29): the same with `get_qc_classifier_compressed` (Haar x 2)."""
    return dsp_qc_flt_optimization(wvfs, config, tau, f_evaluate_qc, ctx, _compressed=True)


def dsp_sg_optimization_compressed(wvfs_wdw: ArrayOfRDWaveforms, wvfs_pre: ArrayOfRDWaveforms, config: DSPConfig, tau: float, pars_filter: dict,
                                   presum_rate: float = 8.0, f_evaluate_qc=None, ctx=None) -> Table:
    """`dsp_sg_optimization_compressed(wvfs_wdw, wvfs_pre, config, tau, pars_filter; presum_rate, f_evaluate_qc)` (reference
    :460-511): baseline, t50 and trapezoid energy from the presummed traces (one launch of `ldsp_sg_grid_run` with an empty
    grid), the current maxima per Savitzky-Golay window length from the windowed traces (baseline = blmean / presum_rate),
    through the functor entry points."""
    from .extractors import get_wvf_maximum
    from .filters import InvCRFilter, SavitzkyGolayFilter, shift_waveform
    rt, ft = pars_filter["trap"]["rt"], pars_filter["trap"]["ft"]
    pre = dsp_qc_flt_optimization(wvfs_pre, config, tau, None, ctx, _trap=(rt, ft))
    w = InvCRFilter(float(tau))(shift_waveform(wvfs_wdw, -pre["blmean"] / float(presum_rate)))
    cw = config.current_window
    grid = list(config.a_grid_wl_sg)
    aoe = torch.stack([get_wvf_maximum(SavitzkyGolayFilter(wl, config.sg_flt_degree, 1)(w), cw.left, cw.right) for wl in grid]) / pre["energy"][None, :]
    res = Table()
    res["aoe"] = aoe
    res["energy"], res["blmean"], res["blslope"], res["t50"] = pre["energy"], pre["blmean"], pre["blslope"], pre["t50"]
    if f_evaluate_qc is None:
        res["qc_label"] = torch.full((len(wvfs_pre),), -1, dtype=torch.int32, device=aoe.device)
    else:   # get_qc_classifier_compressed on the baseline-subtracted presummed traces (:480)
        from .ml_routines import get_qc_classifier_compressed
        res["qc_label"] = torch.as_tensor(get_qc_classifier_compressed(wvfs_pre, f_evaluate_qc, config, ctx)).to(torch.int32)
    return res


def dsp_qdrift_flt_optimization(wvfs: ArrayOfRDWaveforms, blmean: torch.Tensor, config: DSPConfig, tau: float) -> torch.Tensor:
    """`dsp_qdrift_flt_optimization(wvfs, blmean, config, tau)` (reference :72-90), spelled with the functor entry points:
    shift by the given baseline, InvCRFilter, get_t0, get_qdrift."""
    from .filters import InvCRFilter, shift_waveform
    from .routines import get_qdrift, get_t0
    kw = config.kwargs_pars
    w = shift_waveform(wvfs, -blmean)
    w = InvCRFilter(tau)(w)
    t0 = get_t0(w, config.t0_threshold, flt_pars=tuple(kw.t0_flt_pars), mintot=kw.t0_mintot)
    return get_qdrift(w, t0, (config.qdrift_int_length.first, config.qdrift_int_length.last),
                      pol_power=int(kw.int_interpolation_order), sign_est_length=kw.int_interpolation_length)
