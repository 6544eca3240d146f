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
from .routines import (ArrayOfRDWaveforms, Table, dsp_icpc, dsp_sipm, dsp_sipm_compressed, icpc_run, icpc_pz_trap_run, sipm_run, table_columns,
                       get_t0, get_threshold, get_qdrift, get_intracePileUp)
from .filters import (SamplingInfo, smplinfo, fltinstance, rdfilt_, flt_output_length, flt_input_length,
                      flt_output_time_axis, InvCRFilter, IntegratorFilter, TrapezoidalChargeFilter, CUSPChargeFilter,
                      ZACChargeFilter, SavitzkyGolayFilter, DerivativeFilter, HaarAveragingFilter, MovingWindowFilter,
                      MovingWindowMultiFilter, TruncateFilter, TimeAxisFilter, shift_waveform, multiply_waveform, reverse_waveform)
from .optimization import (dsp_trap_rt_optimization, dsp_trap_ft_optimization, dsp_cusp_rt_optimization, dsp_zac_rt_optimization,
                           dsp_cusp_ft_optimization, dsp_zac_ft_optimization, dsp_sg_optimization, dsp_sg_optimization_compressed, dsp_qc_flt_optimization, dsp_qc_flt_optimization_compressed, dsp_qdrift_flt_optimization, trap_grid_run, fir_grid_run, lower_trap_grid,
                           cuspzac_grid_taps)
from .thin_routines import (dsp_decay_times, dsp_puls, dsp_puls_compressed, dsp_pmts, dsp_sg_sipm_thresholds_compressed,
                            dsp_sg_sipm_optimization_compressed)
from .compressed import dsp_icpc_compressed, slope_residual_sigma
from .ml_routines import get_qc_classifier, get_qc_classifier_compressed, qc_features, RbfSvmPredictor
from .extractors import (VectorOfVectors, signalstats, tailstats, extremestats, thresholdstats, thresholdstats_mad,
                         saturation, get_wvf_maximum, Intersect, IntersectMaximum, MultiIntersect, PolynomialDNI,
                         SignalEstimator)

__all__ = [n for n in dir() if not n.startswith("_")]
