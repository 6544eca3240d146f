"""`dsp_icpc_compressed` (reference src/dsp_icpc.jl:293-499): the chain on the pair of traces production
LEGEND data stores per event — `waveform_presummed` (the whole trace summed in groups of `presum_rate` samples: energies,
baseline, tail) and `waveform_windowed` (full sampling rate around the rise: timing, current).

`decode_data` (LegendDataTypes, :319-320) is the I/O side's codec and stays there: both columns are taken as decoded
ArrayOfRDWaveforms.  Two launches of the fused kernel (`ldsp_icpc_run`): the presummed traces with parameters lowered for
their sampling step (rail and SG window scaled by the rate), then the windowed traces — short, with the baseline handed over
from the presummed ones and without the CUSP/ZAC stage (`ldsp_icpc_run_opts`: explicit per-call arguments).  `fused_windowed=False` spells the
windowed half statement by statement through the filter-functor / extractor entry points instead (the comparator).
"""
from __future__ import annotations

import torch

from . import _lib
from .config import DSPConfig, DEFAULT_T0_FLT_PARS, get_fltpars, lower_icpc, window_index
from .extractors import get_wvf_maximum, signalstats
from .filters import DerivativeFilter, InvCRFilter, SavitzkyGolayFilter, multiply_waveform, shift_waveform
from .routines import ArrayOfRDWaveforms, Table, get_qdrift, get_t0, get_threshold, icpc_run, table_columns


def slope_residual_sigma(st: dict, n: int, dt: float) -> torch.Tensor:
    """`signalstats(...).slope_residual_sigma` (read at reference src/dsp_icpc.jl:468-481; RadiationDetectorDSP, source not
    in the container — assumption A8): population standard deviation of the residuals about the fitted line,
    sqrt(var_y - slope^2 var_t), var_t = dt^2 (n^2 - 1) / 12 for n equidistant points."""
    var_t = float(dt) ** 2 * (float(n) ** 2 - 1.0) / 12.0
    v = st["sigma"].double() ** 2 - st["slope"].double() ** 2 * var_t
    return v.clamp_min(0.0).sqrt().float()


WINDOWED_COLS = ("e_max", "e_min", "t0", "t10", "t50", "t80", "t90", "t99", "drift_time", "qdrift", "lq", "a_raw", "a_sg", "a_60", "a_100", "t0_inv")


def windowed_columns(wdw: ArrayOfRDWaveforms, blmean_pre: torch.Tensor, rate: int, config: DSPConfig, tau: float, pars_filter: dict,
                     ctx: _lib.Context = None) -> dict:
    """The columns `dsp_icpc_compressed` takes from the windowed traces (:352-353, :362-393, :431-435, :452-459), by ONE launch
    of the fused kernel: baseline = blmean of the presummed trace / rate, no CUSP/ZAC stage (both arguments of
    `ldsp_icpc_run_opts`), parameters lowered for the window's own time axis."""
    x = wdw.signal
    ctx = ctx or _lib.default_context(x.device.index)
    pw = lower_icpc(config, tau, pars_filter, wdw.nsamples, wdw.t_first, wdw.dt, windowed=True)
    bl = blmean_pre.to(device=x.device, dtype=torch.float32).contiguous()
    B = table_columns(icpc_run(x, pw, ctx, ext_baseline=bl, ext_baseline_scale=1.0 / float(rate), main_only=True))
    return {k: B[k] for k in WINDOWED_COLS}


def windowed_columns_unfused(wdw: ArrayOfRDWaveforms, blmean_pre: torch.Tensor, rate: int, config: DSPConfig, tau: float,
                             pars_filter: dict) -> dict:
    """The same columns statement by statement through the filter-functor / extractor entry points, as the reference spells
    them (the comparator of `windowed_columns` in tests/test_compressed_gpu.py)."""
    kw = config.kwargs_pars
    w = shift_waveform(wdw, -blmean_pre / float(rate))
    wmax, wmin = w.signal.amax(dim=1), w.signal.amin(dim=1)
    w = InvCRFilter(float(tau))(w)
    t0 = get_t0(w, config.t0_threshold, flt_pars=tuple(kw.t0_flt_pars), mintot=kw.t0_mintot)
    tx = {f: get_threshold(w, wmax * f, mintot=kw.tx_mintot) for f in (0.1, 0.5, 0.8, 0.9, 0.99)}
    qd = (config.qdrift_int_length.first, config.qdrift_int_length.last)
    lqr = (config.lq_int_length.first, config.lq_int_length.last)
    cw = config.current_window
    sg_wl = get_fltpars(pars_filter, "sg", config)
    res = dict(e_max=wmax, e_min=wmin, t0=t0, t10=tx[0.1], t50=tx[0.5], t80=tx[0.8], t90=tx[0.9], t99=tx[0.99],
               drift_time=(tx[0.9] - t0) * 1000.0)        # uconvert(ns, t90 - t0)
    res["qdrift"] = get_qdrift(w, t0, qd, pol_power=kw.int_interpolation_order, sign_est_length=kw.int_interpolation_length)
    res["lq"] = get_qdrift(w, tx[0.8], lqr, pol_power=kw.int_interpolation_order, sign_est_length=kw.int_interpolation_length)
    res["a_raw"] = get_wvf_maximum(DerivativeFilter(1.0)(w), cw.left, cw.right)
    for k, wl in (("a_sg", sg_wl), ("a_60", 60.0), ("a_100", 100.0)):
        res[k] = get_wvf_maximum(SavitzkyGolayFilter(wl, config.sg_flt_degree, 1)(w), cw.left, cw.right)
    res["t0_inv"] = get_t0(multiply_waveform(w, -1.0), config.t0_threshold, flt_pars=DEFAULT_T0_FLT_PARS, mintot=kw.t0_mintot)
    return res


def dsp_icpc_compressed(data: Table, config: DSPConfig, tau: float, pars_filter: dict, f_evaluate_qc=None,
                        ctx: _lib.Context = None, fused_windowed: bool = True) -> Table:
    """`dsp_icpc_compressed(data, config, τ, pars_filter; f_evaluate_qc)` — reference src/dsp_icpc.jl:293-499, same column
    names.  Required columns: waveform_presummed, waveform_windowed, presum_rate, baseline, timestamp, eventnumber,
    daqenergy, t_sat_lo, t_sat_hi, deadtime."""
    pre: ArrayOfRDWaveforms = data["waveform_presummed"]
    wdw: ArrayOfRDWaveforms = data["waveform_windowed"]
    rates = torch.unique(torch.as_tensor(data["presum_rate"]))
    if rates.numel() != 1:
        raise ValueError("presum_rate must be the same for every trace (only(unique(presum_rate)), dsp_icpc.jl:330)")
    rate = int(rates[0])
    kw = config.kwargs_pars
    n = len(pre)

    # ---- presummed traces: one launch of the fused chain
    pa = lower_icpc(config, tau, pars_filter, pre.nsamples, pre.t_first, pre.dt, presum_rate=rate)
    A = table_columns(icpc_run(pre.signal, pa, ctx))
    dev = A["blmean"].device
    npts = lambda win: window_index(win.right, pre.t_first, pre.dt) - window_index(win.left, pre.t_first, pre.dt) + 1
    bl_stats = dict(mean=A["blmean"], sigma=A["blsigma"], slope=A["blslope"])
    aux = {}
    for name, win in (("auxbl1", config.auxbl1_window), ("auxbl2", config.auxbl2_window),
                      ("auxpz1", config.auxpz1_window), ("auxpz2", config.auxpz2_window)):
        st = signalstats(pre, win.left, win.right)
        if name.startswith("auxpz"):            # taken after shift_waveform(-blmean), before the deconvolution  :365-366
            st["mean"] = st["mean"] - A["blmean"]
        aux[name] = (st, slope_residual_sigma(st, npts(win), pre.dt))

    # ---- windowed traces: the fused chain again, without its CUSP/ZAC stage, the baseline handed over  (:352-353)
    W = windowed_columns(wdw, A["blmean"], rate, config, tau, pars_filter, ctx) if fused_windowed else \
        windowed_columns_unfused(wdw, A["blmean"], rate, config, tau, pars_filter)
    wmax, wmin, t0, t0_inv, drift_time, qdrift, lq, a_raw = (W[k] for k in ("e_max", "e_min", "t0", "t0_inv", "drift_time", "qdrift", "lq", "a_raw"))
    tx = {0.1: W["t10"], 0.5: W["t50"], 0.8: W["t80"], 0.9: W["t90"], 0.99: W["t99"]}
    a = {k: W[k] for k in ("a_sg", "a_60", "a_100")}

    if f_evaluate_qc is None:
        qc = torch.full((n,), -1, dtype=torch.int64, device=dev)
    else:   # get_qc_classifier_compressed on the baseline-subtracted presummed traces  (:355-356)
        from .ml_routines import get_qc_classifier_compressed
        qc = torch.as_tensor(get_qc_classifier_compressed(pre, f_evaluate_qc, config, ctx)).to(torch.int64)

    res = Table()
    res["blfc"], res["timestamp"], res["eventID_fadc"], res["e_fc"] = data["baseline"], data["timestamp"], data["eventnumber"], data["daqenergy"]
    res["deadtime"] = data["deadtime"]
    for k in ("n_sat_low", "n_sat_high", "n_sat_low_cons", "n_sat_high_cons"):
        res[k] = A[k]
    res["t_sat_lo"], res["t_sat_hi"] = data["t_sat_lo"], data["t_sat_hi"]
    for k in ("blmean", "blsigma", "blslope", "bloffset"):
        res[k] = A[k]
    res["bl_slope_sigma"] = slope_residual_sigma(bl_stats, npts(config.bl_window), pre.dt)
    for name in ("auxbl1", "auxbl2"):
        st, srs = aux[name]
        res[f"{name}_mean"], res[f"{name}_sigma"], res[f"{name}_slope_sigma"] = st["mean"], st["sigma"], srs
    res["qc_label"] = qc
    res["e_max"], res["e_min"], res["e_max_pre"], res["e_min_pre"] = wmax, wmin, A["e_max"], A["e_min"]
    for k in ("tailmean", "tailsigma", "tailslope", "tailoffset"):
        res[k] = A[k]
    res["tail_τ"], res["tail_mean"], res["tail_sigma"] = A["tail_tau"], A["tail_mean"], A["tail_sigma"]
    for name in ("auxpz1", "auxpz2"):
        st, srs = aux[name]
        res[f"{name}_mean"], res[f"{name}_sigma"], res[f"{name}_slope_sigma"] = st["mean"], st["sigma"], srs
    res["t0"], res["t10"], res["t50"], res["t80"], res["t90"], res["t99"] = t0, tx[0.1], tx[0.5], tx[0.8], tx[0.9], tx[0.99]
    res["t50_pre"] = A["t50"]
    res["drift_time"], res["t50_current"] = drift_time, A["t50_current"]
    for k in ("e_10410", "e_535", "e_313", "e_trap", "e_cusp", "e_zac", "e_trap_max", "e_cusp_max", "e_zac_max",
              "t_trap_max", "t_cusp_max", "t_zac_max"):
        res[k] = A[k]
    res["qdrift"], res["lq"] = qdrift, lq
    res["a_sg"], res["a_60"], res["a_100"], res["a_raw"] = a["a_sg"], a["a_60"], a["a_100"], a_raw
    res["inTrace_intersect"], res["inTrace_n"] = A["inTrace_intersect"], A["inTrace_n"]
    res["e_10410_inv"], res["e_313_inv"], res["t0_inv"] = A["e_10410_inv"], A["e_313_inv"], t0_inv
    return res
