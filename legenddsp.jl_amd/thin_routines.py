"""Thin recombinations of the hot path's stages: `dsp_decay_times` (reference src/dsp_decaytime.jl:11-26),
`dsp_puls` (src/dsp_puls.jl:29-65) and `dsp_pmts` (src/dsp_pmts.jl:3-65), spelled with the filter-functor / extractor
entry points."""
from __future__ import annotations

import torch

from .config import DSPConfig
from .extractors import IntersectMaximum, extremestats, saturation, signalstats, tailstats
from .filters import SavitzkyGolayFilter, TimeAxisFilter, TrapezoidalChargeFilter, shift_waveform
from .routines import ArrayOfRDWaveforms, Table, get_threshold

_US = 1000.0  # ns per us


def dsp_decay_times(wvfs: ArrayOfRDWaveforms, bl_window_or_config, tail_window=None) -> torch.Tensor:
    """`dsp_decay_times(wvfs, bl_window, tail_window)` / `dsp_decay_times(wvfs, config)`: decay time of the tail in us."""
    if isinstance(bl_window_or_config, DSPConfig):
        bl, tail = bl_window_or_config.bl_window, bl_window_or_config.tail_window
        bl, tail = (bl.left, bl.right), (tail.left, tail.right)
    else:
        bl, tail = tuple(bl_window_or_config), tuple(tail_window)
    st = signalstats(wvfs, bl[0], bl[1])
    w = shift_waveform(wvfs, -st["mean"])
    return tailstats(w, tail[0], tail[1])["τ"] / _US


def dsp_puls(data: Table, config: DSPConfig, _waveform_column="waveform") -> Table:
    """`dsp_puls(data, config)`: baseline statistics, t50 at half maximum, maximum, Trap(10 us, 4 us) maximum of the
    baseline-subtracted pulser traces + the four passthrough columns."""
    wvfs: ArrayOfRDWaveforms = data[_waveform_column]
    bl = config.bl_window
    st = signalstats(wvfs, bl.left, bl.right)
    w = shift_waveform(wvfs, -st["mean"])
    wvf_max = w.signal.max(dim=1).values
    t50 = get_threshold(w, 0.5 * wvf_max)
    e_10410 = TrapezoidalChargeFilter(10 * _US, 4 * _US)(w).signal.max(dim=1).values
    res = Table()
    res["blmean"], res["blsigma"], res["blslope"], res["bloffset"] = st["mean"], st["sigma"], st["slope"], st["offset"]
    res["t50"], res["e_max"], res["e_10410"] = t50, wvf_max, e_10410
    res["blfc"], res["timestamp"], res["eventID_fadc"], res["e_fc"] = data["baseline"], data["timestamp"], data["eventnumber"], data["daqenergy"]
    return res


def dsp_pmts(data: Table, config: dict) -> Table:
    """`dsp_pmts(data, config)` — reference src/dsp_pmts.jl:3-65.  `decode_data` (:20) is the I/O side's codec: the
    `waveform` column is taken decoded.  `TimeAxisFilter(time_axis_step_length)` (src/timeaxis.jl:31-60) only replaces the
    sampling step of the time axis.  `wsg_weight != 0` selects the WeightedSavitzkyGolayFilter of
    src/alternative_filters.jl, which is outside the hot path: it raises."""
    cfg = config
    if int(cfg["wsg_weight"]) != 0:
        raise NotImplementedError("WeightedSavitzkyGolayFilter (reference src/alternative_filters.jl) is out of scope; wsg_weight = 0 "
                                  "selects the plain SavitzkyGolayFilter (src/dsp_pmts.jl:44-45)")
    w0: ArrayOfRDWaveforms = data["waveform"]
    wvfs = TimeAxisFilter(float(cfg["time_axis_step_length"]))(w0)   # src/dsp_pmts.jl:23
    bl = signalstats(wvfs, cfg["baseline_window_start"], cfg["baseline_window_end"])
    wf_blsub = shift_waveform(wvfs, -bl["mean"])
    raw = extremestats(wf_blsub)
    trig = IntersectMaximum(cfg["min_tot_intersect"], cfg["max_tot_intersect"])(wf_blsub, float(cfg["intersect_threshold"]))
    sat = saturation(wvfs, cfg["saturation_limit_low"], cfg["saturation_limit_high"])
    pulse = extremestats(SavitzkyGolayFilter(cfg["wsg_window_length"], int(cfg["wsg_flt_degree"]), 0)(wf_blsub))
    res = Table()
    res["timestamp"], res["eventID_fadc"], res["e_fc"], res["channel"] = data["timestamp"], data["eventnumber"], data["daqenergy"], data["channel"]
    res["raw_pulse_height"], res["raw_pulse_low"], res["raw_t0_hi"], res["raw_t0_low"] = raw["max"], raw["min"], raw["tmax"], raw["tmin"]
    res["trig_max"], res["trig_t"], res["trig_mult"] = trig["max"], trig["x"], trig["multiplicity"]
    res["sat_low"], res["sat_high"] = sat["low"], sat["high"]
    res["pulse_height"], res["pulse_low"], res["t0_hi"], res["t0_low"] = pulse["max"], pulse["min"], pulse["tmax"], pulse["tmin"]
    res["bl_mean"], res["bl_sigma"], res["bl_slope"] = bl["mean"], bl["sigma"], bl["slope"]
    return res


def dsp_puls_compressed(data: Table, config: DSPConfig) -> Table:
    """`dsp_puls_compressed(data, config)` — reference src/dsp_puls.jl:98-134: `dsp_puls` on the decoded `waveform_presummed`."""
    return dsp_puls(data, config, _waveform_column="waveform_presummed")


# ---- SiPM threshold / window-length scans (reference src/dsp_sipm_optimization.jl).  The per-trace filters and trigger
# finders are the HIP entry points; what is combined ACROSS traces (one masked standard deviation over the pooled samples of
# the first n traces) is a device reduction by torch — a batch statistic, not part of the per-waveform path.

def dsp_sg_sipm_thresholds_compressed(wvfs: ArrayOfRDWaveforms, sg_window_length: float, config: dict) -> Table:
    """`dsp_sg_sipm_thresholds_compressed(wvfs, sg_window_length, config)` (:16-47): the pooled samples of the SG derivative and
    of its integral (and the latter flipped)."""
    from .filters import IntegratorFilter
    sg = SavitzkyGolayFilter(sg_window_length, int(config["sg_flt_degree"]), 1)(wvfs)
    integ = IntegratorFilter(1.0)(sg)
    res = Table()
    res["bsl_deriv"] = sg.signal.reshape(-1)
    res["bsl"] = integ.signal.reshape(-1)
    res["bsl_flipped"] = -res["bsl"]
    return res


def _pooled_thresholdstats(v: torch.Tensor, lo: float, hi: float) -> float:
    """`thresholdstats(bsl, min, max)` (src/thresholdstats.jl:14-41) of one pooled vector: population sigma of lo <= v <= hi."""
    m = (v >= lo) & (v <= hi)
    k = m.sum().double()
    vv = torch.where(m, v, torch.zeros_like(v)).double()
    mean = vv.sum() / k
    return float(torch.sqrt(torch.clamp((vv * vv).sum() / k - mean * mean, min=0.0)))


def dsp_sg_sipm_optimization_compressed(wvfs: ArrayOfRDWaveforms, dsp_config: dict, optimization_config: dict, n_max_wvfs: int = None) -> Table:
    """`dsp_sg_sipm_optimization_compressed([n_max_wvfs,] wvfs, dsp_config, optimization_config)` (:68-135): per SG window
    length of `e_grid_wl`, the trigger maxima of all traces at the threshold n_sigma x (masked sigma of the first n_wvfs
    traces' pooled samples); with `n_max_wvfs` the batch is processed in partitions and the smallest threshold per window
    length is reported, as the reference's partitioned method does."""
    from .extractors import VectorOfVectors
    if n_max_wvfs is not None:
        parts = [dsp_sg_sipm_optimization_compressed(ArrayOfRDWaveforms(wvfs.signal[a:a + n_max_wvfs], wvfs.t_first, wvfs.dt), dsp_config,
                                                     optimization_config) for a in range(0, len(wvfs), n_max_wvfs)]
        W = len(parts[0]["thresholds_grid"])
        vals = [torch.cat([p["trig_max_grid"][w] for p in parts]) for w in range(W)]
        thr = [min(p["thresholds_grid"][w] for p in parts) for w in range(W)]
    else:
        thr_cfg = optimization_config["threshold"]
        n_thr = min(len(wvfs), int(thr_cfg["n_wvfs"]))
        vals, thr = [], []
        for wl in optimization_config["e_grid_wl"]:
            sg = SavitzkyGolayFilter(wl, int(dsp_config["sg_flt_degree"]), 1)(wvfs)
            t = _pooled_thresholdstats(sg.signal[:n_thr].reshape(-1), float(thr_cfg["min_cut"]), float(thr_cfg["max_cut"])) * float(dsp_config["n_σ_threshold"])
            inters = IntersectMaximum(dsp_config["min_tot_intersect"], dsp_config["max_tot_intersect"])(sg, t)
            vals.append(inters["max"].values); thr.append(t)
    offs = torch.zeros(len(vals) + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(torch.tensor([len(v) for v in vals]), 0)
    res = Table()
    res["trig_max_grid"] = VectorOfVectors(offs.to(vals[0].device), torch.cat(vals))
    res["thresholds_grid"] = thr
    return res
