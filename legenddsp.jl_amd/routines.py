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


def icpc_run(wf: torch.Tensor, params: _abi.IcpcParams, ctx: _lib.Context = None, out: torch.Tensor = None) -> torch.Tensor:
    """Run the fused kernel on a device-resident [n, L] float32 batch.

    Returns the [n, 48] float32 output table (columns in `_abi.ICPC_COLS` order;
    the 5 integer columns hold int32 bit patterns — see `table_columns`)."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "icpc_run needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if L != params.L:
        raise ValueError(f"waveform length {L} != params.L {params.L}")
    wf = _as_device_f32(wf, wf.device)
    nc = len(_abi.ICPC_COLS)
    if out is None:
        out = torch.empty((n, nc), dtype=torch.float32, device=wf.device)
    assert out.shape == (n, nc) and out.is_contiguous() and out.dtype == torch.float32
    o = _abi.IcpcOut()
    base = out.data_ptr()
    for i, c in enumerate(_abi.ICPC_COLS):
        setattr(o, c, base + 4 * i)
    o.stride = nc
    ctx.bind_stream()
    _lib.check(_lib.lib().ldsp_icpc_run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params), C.byref(o)))
    return out


def icpc_pz_trap_run(wf: torch.Tensor, params: _abi.IcpcParams, ctx: _lib.Context = None, out: torch.Tensor = None):
    """BASELINE config 2 sub-chain (blmean, e_10410) — `ldsp_icpc_pz_trap_run`. Returns a [2, n] tensor."""
    if not wf.is_cuda:
        raise _lib.LdspError(-103, "needs a device-resident waveform tensor (no CPU fallback)")
    ctx = ctx or _lib.default_context(wf.device.index)
    n, L = wf.shape
    if out is None:
        out = torch.empty((2, n), dtype=torch.float32, device=wf.device)
    ctx.bind_stream()
    _lib.check(_lib.lib().ldsp_icpc_pz_trap_run(ctx.handle, C.c_void_p(wf.data_ptr()), n, C.byref(params),
                                                C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr())))
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
    if f_evaluate_qc is not None:
        raise NotImplementedError("QC classifier (reference src/dsp_ml_routines.jl) is outside the hot path; "
                                  "qc_label is -1 as with f_evaluate_qc = missing (dsp_icpc.jl:108)")
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
    res["qc_label"] = torch.full((n,), -1, dtype=torch.int64, device=dev)  # dsp_icpc.jl:108
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
