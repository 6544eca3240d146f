"""Thin recombinations of the hot path's stages: `dsp_decay_times` (reference src/dsp_decaytime.jl:11-26)
and `dsp_puls` (src/dsp_puls.jl:29-65), spelled with the filter-functor / extractor entry points."""
from __future__ import annotations

import torch

from .config import DSPConfig
from .extractors import signalstats, tailstats
from .filters import TrapezoidalChargeFilter, shift_waveform
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


def dsp_puls(data: Table, config: DSPConfig) -> Table:
    """`dsp_puls(data, config)`: baseline statistics, t50 at half maximum, maximum, Trap(10 us, 4 us) maximum of the
    baseline-subtracted pulser traces + the four passthrough columns."""
    wvfs: ArrayOfRDWaveforms = data["waveform"]
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
