"""ctypes/numpy front end of oracle/libldsp_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of ldsp_oracle.c for the pinning status).
"""
import ctypes as C
import os
import subprocess

import numpy as np

import legenddsp_jl_amd as _pkg
from legenddsp_jl_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libldsp_oracle.so")
_SO32 = os.path.join(_HERE, "libldsp_oracle_f32.so")   # -DORC_F32: the reference's typing for Float32 input (dsp_icpc only)


def build(force=False):
    src = os.path.join(_HERE, "ldsp_oracle.c")
    if force or not os.path.exists(_SO) or not os.path.exists(_SO32) or min(os.path.getmtime(_SO), os.path.getmtime(_SO32)) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None
_lib32 = None


def lib32():
    """The Float32-typed build (see the `real` typedef in ldsp_oracle.c): only orc_dsp_icpc is meant to be called through it —
    the functor-level entry points of that build take float arrays."""
    global _lib32
    if _lib32 is None:
        if not os.path.exists(_SO32):
            build()
        _lib32 = C.CDLL(_SO32)
    return _lib32


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_thresholdstats.restype = C.c_double
        _lib.orc_thresholdstats_mad.restype = C.c_double
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    pass


def _chk(rc):
    if rc < 0:
        raise OracleError(f"oracle error {rc}")
    return rc


# ---- extractors -------------------------------------------------------------

def signalstats(y, frm, until, t_first=0.0, dt=1.0):
    y = _d(y); o = (C.c_double * 4)()
    _chk(lib().orc_signalstats(_p(y), len(y), int(frm), int(until), C.c_double(t_first), C.c_double(dt),
                               C.byref(o, 0), C.byref(o, 8), C.byref(o, 16), C.byref(o, 24)))
    return dict(mean=o[0], sigma=o[1], slope=o[2], offset=o[3])


def tailstats(y, frm, until, t_first=0.0, dt=1.0):
    y = _d(y); o = (C.c_double * 3)()
    _chk(lib().orc_tailstats(_p(y), len(y), int(frm), int(until), C.c_double(t_first), C.c_double(dt),
                             C.byref(o, 0), C.byref(o, 8), C.byref(o, 16)))
    return dict(mean=o[0], sigma=o[1], tau=o[2])


def extremestats(y, frm=None, until=None, t_first=0.0, dt=1.0):
    y = _d(y); o = (C.c_double * 4)()
    frm = 0 if frm is None else frm
    until = len(y) - 1 if until is None else until
    _chk(lib().orc_extremestats(_p(y), len(y), int(frm), int(until), C.c_double(t_first), C.c_double(dt),
                                C.byref(o, 0), C.byref(o, 8), C.byref(o, 16), C.byref(o, 24)))
    return dict(min=o[0], max=o[1], tmin=o[2], tmax=o[3])


def thresholdstats(y, lo=-np.inf, hi=np.inf):
    y = _d(y)
    return lib().orc_thresholdstats(_p(y), len(y), C.c_double(lo), C.c_double(hi))


def thresholdstats_mad(y, lo=-np.inf, hi=np.inf):
    y = _d(y)
    return lib().orc_thresholdstats_mad(_p(y), len(y), C.c_double(lo), C.c_double(hi))


def saturation(y, low, high, frm=None, until=None):
    y = _d(y); o = (C.c_int * 4)()
    frm = 0 if frm is None else frm
    until = len(y) - 1 if until is None else until
    _chk(lib().orc_saturation(_p(y), len(y), int(frm), int(until), C.c_double(low), C.c_double(high), o))
    return dict(low=o[0], high=o[1], max_cons_low=o[2], max_cons_high=o[3])


def get_wvf_maximum(y, frm, until):
    y = _d(y); o = C.c_double()
    _chk(lib().orc_get_wvf_maximum(_p(y), len(y), int(frm), int(until), C.byref(o)))
    return o.value


def intersect(y, thr, min_n, t_first=0.0, dt=1.0):
    y = _d(y); x = C.c_double(); m = C.c_int()
    lib().orc_intersect(_p(y), len(y), C.c_double(t_first), C.c_double(dt), C.c_double(thr), int(min_n),
                        C.byref(x), C.byref(m))
    return dict(x=x.value, multiplicity=m.value)


def intersect_maximum(y, thr, min_n, max_n, t_first=0.0, dt=1.0, cap=None):
    y = _d(y)
    cap = max(1, len(y)) if cap is None else cap
    bufs = [np.full(cap, np.nan) for _ in range(4)]
    n = lib().orc_intersect_maximum(_p(y), len(y), C.c_double(t_first), C.c_double(dt), C.c_double(thr),
                                    int(min_n), int(max_n), int(cap), *[_p(b) for b in bufs])
    k = min(n, cap)
    return dict(x=bufs[0][:k], x_high=bufs[1][:k], x_tot=bufs[2][:k], max=bufs[3][:k], multiplicity=n)


def multi_intersect(y, ratios, min_n, half_n=1, degree=1, rate=1, t_first=0.0, dt=1.0):
    y = _d(y); r = _d(ratios); out = np.zeros(len(r))
    _chk(lib().orc_multi_intersect(_p(y), len(y), C.c_double(t_first), C.c_double(dt), _p(r), len(r),
                                   int(min_n), int(half_n), int(degree), int(rate), _p(out)))
    return out


def signal_estimator(y, t, npts, degree, t_first=0.0, dt=1.0):
    y = _d(y); o = C.c_double()
    _chk(lib().orc_signal_estimator(_p(y), len(y), C.c_double(t_first), C.c_double(dt), C.c_double(t),
                                    int(npts), int(degree), C.byref(o)))
    return o.value


# ---- filters ----------------------------------------------------------------

def _flt(fn, x, *args, nout=None):
    x = _d(x); y = np.empty(len(x) if nout is None else nout)
    n = _chk(fn(_p(x), len(x), *args, _p(y)))
    return y[:n]


def invcr(x, c):
    return _flt(lib().orc_invcr, x, C.c_double(c))


def integrator(x, gain=1.0):
    return _flt(lib().orc_integrator, x, C.c_double(gain))


def trap(x, navg, ngap, navg2=None):
    return _flt(lib().orc_trap, x, int(navg), int(ngap), int(navg if navg2 is None else navg2))


def fir(x, h):
    h = _d(h)
    return _flt(lib().orc_fir, x, _p(h), len(h))


def derivative(x, gain=1.0):
    return _flt(lib().orc_derivative, x, C.c_double(gain))


def haar(x, ds):
    return _flt(lib().orc_haar, x, int(ds))


def moving_window(x, l):
    return _flt(lib().orc_moving_window, x, int(l))


def moving_window_multi(x, l):
    return _flt(lib().orc_moving_window_multi, x, int(l))


def cusp_coeffs(p: _abi.CuspZac):
    h = np.empty(p.length); _chk(lib().orc_cusp_coeffs(C.byref(p), _p(h))); return h


def zac_coeffs(p: _abi.CuspZac):
    h = np.empty(p.length); _chk(lib().orc_zac_coeffs(C.byref(p), _p(h))); return h


def sg_coeffs(npts, degree, deriv):
    h = np.empty(npts); _chk(lib().orc_sg_coeffs(int(npts), int(degree), int(deriv), _p(h))); return h


# ---- fused routines -----------------------------------------------------------

def set_fir_mode(fft: bool):
    """CPU-baseline leg only: evaluate the long FIR filters (CUSP / ZAC) by FFT instead of direct convolution (ldsp_oracle.c)."""
    lib().orc_set_fir_mode(1 if fft else 0)


def dsp_icpc(wf, params: _abi.IcpcParams, nthreads=1, strict=True, f32=False):
    """wf: [n][L] float32 -> dict of float64 columns in _abi.ICPC_COLS order.  f32: the Float32-typed restatement (what the
    reference computes for Float32 input), for the per-column float32 envelope of tests/parity.py."""
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    n, L = wf.shape
    assert L == params.L
    L_ = lib32() if f32 else lib()
    nc = L_.orc_icpc_ncols()
    assert nc == len(_abi.ICPC_COLS)
    out = np.empty((n, nc)); status = np.zeros(n, dtype=np.int32)
    rc = L_.orc_dsp_icpc(_p(wf), C.c_long(n), C.byref(params), _p(out), _p(status), int(nthreads))
    if strict:
        _chk(rc)
    cols = {c: out[:, i].copy() for i, c in enumerate(_abi.ICPC_COLS)}
    cols["_status"] = status
    return cols


def icpc_pz_trap(wf, params: _abi.IcpcParams):
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    n, L = wf.shape
    out = np.empty((n, 2))
    _chk(lib().orc_icpc_pz_trap(_p(wf), C.c_long(n), C.byref(params), _p(out)))
    return dict(blmean=out[:, 0].copy(), e_10410=out[:, 1].copy())


def dsp_sipm(wf, params: _abi.SipmParams, cap=_abi.LDSP_MAX_TRIG, nthreads=1):
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    n, L = wf.shape
    assert L == params.L
    ns_ = lib().orc_sipm_ncols()
    assert ns_ == len(_abi.SIPM_SCALAR_COLS)
    sc = np.empty((n, ns_)); tr = np.empty((n, 16, cap)); cnt = np.zeros((n, 4), dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    _chk(lib().orc_dsp_sipm(_p(wf), C.c_long(n), C.byref(params), _p(sc), _p(tr), _p(cnt), int(cap), _p(status),
                            int(nthreads)))
    res = {c: sc[:, i].copy() for i, c in enumerate(_abi.SIPM_SCALAR_COLS)}
    for g, name in enumerate(_abi.SIPM_TRIG_GROUPS):
        res[name] = dict(count=cnt[:, g].copy(), x=tr[:, 4 * g + 0], x_high=tr[:, 4 * g + 1],
                         x_tot=tr[:, 4 * g + 2], max=tr[:, 4 * g + 3])
    return res


def trap_grid(wf, params, traps, offsets=None):
    """CPU restatement of ldsp_trap_grid_run (reference src/dsp_filter_optimization.jl:102-133, :241-274), composed
    from the functor restatements above, trace by trace: signalstats mean -> shift -> InvCR -> [t50] ->
    per grid point TrapezoidalChargeFilter -> SignalEstimator at the pick-off.  Returns [G, n] float64."""
    wf = np.asarray(wf, dtype=np.float64)
    n, L = wf.shape
    t0, dt = params.t_first, params.dt
    out = np.empty((len(traps), n))
    for i in range(n):
        x = wf[i] - signalstats(wf[i], params.bl_from, params.bl_until, t0, dt)["mean"]
        y = invcr(x, params.pz_c)
        if params.pick_mode == 1:
            r = intersect(y, 0.5 * y.max(), params.tx_mintot, t0, dt)
            t50 = 0.0 if np.isnan(r["x"]) else r["x"]      # get_threshold: NaN -> 0 (dsp_routines.jl:41)
        for g, tr in enumerate(traps):
            f = trap(y, tr.navg, tr.ngap, tr.navg2)
            flen = tr.navg + tr.ngap + tr.navg2
            tf = t0 + dt * (flen - 1)                        # trailing alignment (A1)
            t = params.pick_time if params.pick_mode == 0 else t50 + offsets[g]
            out[g, i] = signal_estimator(f, t, params.sig_est.npts, params.sig_est.degree, tf, dt)
    return out


def fir_grid(wf, params, taps, offsets=None):
    """CPU restatement of ldsp_fir_grid_run (reference src/dsp_filter_optimization.jl:145-229, 286-374): as trap_grid
    with an arbitrary FIR per grid point (taps [G, Lf], FIR order, valid mode, trailing time axis)."""
    wf = np.asarray(wf, dtype=np.float64)
    taps = np.asarray(taps, dtype=np.float64)
    n, L = wf.shape
    t0, dt = params.t_first, params.dt
    G, Lf = taps.shape
    out = np.empty((G, n))
    for i in range(n):
        x = wf[i] - signalstats(wf[i], params.bl_from, params.bl_until, t0, dt)["mean"]
        y = invcr(x, params.pz_c)
        if params.pick_mode == 1:
            r = intersect(y, 0.5 * y.max(), params.tx_mintot, t0, dt)
            t50 = 0.0 if np.isnan(r["x"]) else r["x"]
        for g in range(G):
            f = fir(y, taps[g])
            tf = t0 + dt * (Lf - 1)
            t = params.pick_time if params.pick_mode == 0 else t50 + offsets[g]
            out[g, i] = signal_estimator(f, t, params.sig_est.npts, params.sig_est.degree, tf, dt)
    return out


def sg_optimization(wf, params, trap_, trap_offset, npts, degree, frm, until):
    """CPU restatement of ldsp_sg_grid_run (reference src/dsp_filter_optimization.jl:393-441).  Returns a dict with
    amax [W, n], energy, t50_us, blmean, blslope."""
    wf = np.asarray(wf, dtype=np.float64)
    n, L = wf.shape
    t0, dt = params.t_first, params.dt
    W = len(npts)
    out = dict(amax=np.empty((W, n)), energy=np.empty(n), t50_us=np.empty(n), blmean=np.empty(n), blslope=np.empty(n))
    for i in range(n):
        st = signalstats(wf[i], params.bl_from, params.bl_until, t0, dt)
        y = invcr(wf[i] - st["mean"], params.pz_c)
        r = intersect(y, 0.5 * y.max(), params.tx_mintot, t0, dt)
        t50 = 0.0 if np.isnan(r["x"]) else r["x"]
        f = trap(y, trap_.navg, trap_.ngap, trap_.navg2)
        flen = trap_.navg + trap_.ngap + trap_.navg2
        out["energy"][i] = signal_estimator(f, t50 + trap_offset, params.sig_est.npts, params.sig_est.degree, t0 + dt * (flen - 1), dt)
        out["t50_us"][i] = t50 / 1000.0
        out["blmean"][i], out["blslope"][i] = st["mean"], st["slope"]
        for g in range(W):
            gsg = fir(y, sg_coeffs(int(npts[g]), degree, 1))
            out["amax"][g, i] = get_wvf_maximum(gsg, int(frm[g]), int(until[g]))
    return out


def qc_features(wf, levels, bl_from=-1, bl_until=-1):
    """get_qc_classifier front end (reference src/dsp_ml_routines.jl:9-34, 45-70) in float64: optional
    signalstats(bl).mean + shift, HaarAveragingFilter(2) `levels` times, division by max(|min|, |max|) (0 -> 1).
    Returns (features [n][Lout], norm [n])."""
    wf = np.asarray(wf, dtype=np.float64)
    rows, norms = [], []
    for x in wf:
        if bl_from >= 0:
            x = x - signalstats(x, bl_from, bl_until)["mean"]
        for _ in range(int(levels)):
            x = haar(x, 2)
        nf = max(abs(x.min()), abs(x.max()))
        if nf == 0.0:
            nf = 1.0
        rows.append(x * (1.0 / nf)); norms.append(nf)
    return np.stack(rows), np.array(norms)
