"""legenddsp.jl_amd — MI355X-native implementation of the LegendDSP.jl
dsp_icpc / dsp_sipm per-waveform filter-chain hot path.

Host side (this package) mirrors the reference's operator interface — filter
functors, extractors, DSPConfig, dsp_icpc / dsp_sipm — on top of the C ABI of
`csrc/libldsp_hip.so` (include/ldsp.h).  There is no CPU fallback: every
compute entry point raises if the HIP library is missing.
"""
from . import _abi, _lib, synth
from ._lib import Context, LdspError, build, default_context
from .config import (DSPConfig, PropDict, ClosedInterval, StepRange, get_fltpars, lower_icpc, lower_sipm,
                     reference_test_icpc_config, reference_test_sipm_config, plumbing_icpc_config_4096,
                     ns, us, ms, WindowError)
from .routines import ArrayOfRDWaveforms, Table, dsp_icpc, icpc_run, icpc_pz_trap_run, table_columns

__all__ = [n for n in dir() if not n.startswith("_")]
