"""legenddsp.jl_amd — MI355X-native implementation of the LegendDSP.jl
dsp_icpc / dsp_sipm per-waveform filter-chain hot path.

Host side (this package) mirrors the reference's operator interface — filter
functors, extractors, DSPConfig, dsp_icpc / dsp_sipm — on top of the C ABI of
`csrc/libldsp_hip.so` (include/ldsp.h).  There is no CPU fallback: every
compute entry point raises if the HIP library is missing.
"""
from . import _abi
from .config import (DSPConfig, PropDict, ClosedInterval, StepRange, get_fltpars, lower_icpc, lower_sipm,
                     reference_test_icpc_config, reference_test_sipm_config, ns, us, ms, WindowError)

__all__ = ["DSPConfig", "PropDict", "ClosedInterval", "StepRange", "get_fltpars", "lower_icpc", "lower_sipm",
           "reference_test_icpc_config", "reference_test_sipm_config", "ns", "us", "ms", "WindowError"]
