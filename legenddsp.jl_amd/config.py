"""Config surface of the hot path and its lowering to the POD blocks of ldsp.h.

Mirrors, in Python (no Julia toolchain in the image — DESIGN.md §boundary):
  * `DSPConfig{T}` and `DSPConfig(pd::PropDict)`   reference src/types.jl:32-99,
    src/utils.jl:14-70
  * `get_fltpars(pd, flt, config)`                 reference src/utils.jl:72-82
  * the raw-PropDict config of `dsp_sipm`          reference src/dsp_sipm.jl:49-78

Time quantities (Unitful in the reference) are plain floats in NANOSECONDS, the
time-axis unit of LEGEND waveforms; use the `ns`, `us`, `ms` constants:
`39.0 * us`.  Lowering converts every time to a sample count exactly the way
Julia's `round(Int, t/dt)` does — round half to EVEN — on exact
rationals, so 39 us / 16 ns = 2437.5 -> 2438 and 5 us / 16 ns = 312.5 -> 312.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from fractions import Fraction
from typing import Any

from . import _abi

ns = 1.0
us = 1000.0
ms = 1.0e6
s = 1.0e9
UNIT_PER_US = 1000.0


class PropDict(dict):
    """Nested dict with attribute access (PropDicts.PropDict)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, PropDict):
                self[k] = PropDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


@dataclass(frozen=True)
class ClosedInterval:
    """IntervalSets `a..b`."""
    left: float
    right: float


@dataclass(frozen=True)
class StepRange:
    """`start:step:stop` (only first/last/step are used on the hot path)."""
    start: float
    step: float
    stop: float

    @property
    def first(self):
        return self.start

    @property
    def last(self):
        n = math.floor((self.stop - self.start) / self.step + 1e-9)
        return self.start + n * self.step

    def __len__(self):
        return int(math.floor((self.stop - self.start) / self.step + 1e-9)) + 1

    def __iter__(self):   # the grid scans (dsp_filter_optimization.jl) enumerate the range
        return (self.start + i * self.step for i in range(len(self)))


def _interval(x) -> ClosedInterval:
    if isinstance(x, ClosedInterval):
        return x
    if isinstance(x, dict):
        return ClosedInterval(float(x["min"]), float(x["max"]))
    a, b = x
    return ClosedInterval(float(a), float(b))


def _range(x) -> StepRange:
    if isinstance(x, StepRange):
        return x
    if isinstance(x, dict):
        return StepRange(float(x["start"]), float(x["step"]), float(x["stop"]))
    if len(x) == 3:
        return StepRange(float(x[0]), float(x[1]), float(x[2]))
    # a (first, last) pair as LEGEND metadata stores qdrift_int_length; step 0.1 us (utils.jl:40-42)
    return StepRange(float(x[0]), 0.1 * us, float(x[1]))


@dataclass
class DSPConfig:
    """The 26 fields of reference `DSPConfig{T}` (src/types.jl:32-93)."""
    enc_pickoff_trap: float
    enc_pickoff_zac: float
    enc_pickoff_cusp: float
    flt_length_cusp: float
    flt_length_zac: float
    t0_threshold: float
    inTraceCut_std_threshold: float
    sg_flt_degree: int
    bl_window: ClosedInterval
    tail_window: ClosedInterval
    current_window: ClosedInterval
    qdrift_int_length: StepRange
    lq_int_length: StepRange
    e_grid_rt_trap: StepRange
    e_grid_ft_trap: StepRange
    e_grid_rt_zac: StepRange
    e_grid_ft_zac: StepRange
    e_grid_rt_cusp: StepRange
    e_grid_ft_cusp: StepRange
    a_grid_wl_sg: StepRange
    default_flt_param: PropDict
    kwargs_pars: PropDict
    auxbl1_window: ClosedInterval
    auxbl2_window: ClosedInterval
    auxpz1_window: ClosedInterval
    auxpz2_window: ClosedInterval

    @classmethod
    def from_propdict(cls, pd: dict) -> "DSPConfig":
        """`DSPConfig(pd::PropDict)` — field-by-field copy, src/utils.jl:14-70."""
        pd = PropDict(pd)
        return cls(
            pd.enc_pickoff_trap, pd.enc_pickoff_zac, pd.enc_pickoff_cusp,
            pd.flt_length_cusp, pd.flt_length_zac,
            float(pd.t0_threshold), float(pd.inTraceCut_std_threshold), int(pd.sg_flt_degree),
            _interval(pd.bl_window), _interval(pd.tail_window), _interval(pd.current_window),
            _range(pd.qdrift_int_length), _range(pd.lq_int_length),
            _range(pd.e_grid_trap.rt), _range(pd.e_grid_trap.ft),
            _range(pd.e_grid_zac.rt), _range(pd.e_grid_zac.ft),
            _range(pd.e_grid_cusp.rt), _range(pd.e_grid_cusp.ft),
            _range(pd.a_grid_wl_sg),
            PropDict(pd.flt_defaults), PropDict(pd.kwargs_pars),
            _interval(pd.auxbl1_window), _interval(pd.auxbl2_window),
            _interval(pd.auxpz1_window), _interval(pd.auxpz2_window),
        )


def get_fltpars(pd: dict, flt: str, dspconfig: DSPConfig):
    """reference src/utils.jl:72-82 (per-key fallback to config.default_flt_param)."""
    pd = PropDict(pd)
    if flt == "sg":
        return pd.get(flt, PropDict()).get("wl", dspconfig.default_flt_param[flt])
    if flt not in pd:
        return dspconfig.default_flt_param[flt].rt, dspconfig.default_flt_param[flt].ft
    return (pd[flt].get("rt", dspconfig.default_flt_param[flt].rt),
            pd[flt].get("ft", dspconfig.default_flt_param[flt].ft))


# ---------------------------------------------------------------------------
# lowering helpers

def _frac(x) -> Fraction:
    return Fraction(repr(float(x)))


def round_half_even(x) -> int:
    """Julia `round(Int, x)` on an exact rational."""
    x = x if isinstance(x, Fraction) else _frac(x)
    fl = math.floor(x)
    r = x - fl
    if r > Fraction(1, 2):
        return fl + 1
    if r < Fraction(1, 2):
        return fl
    return fl if fl % 2 == 0 else fl + 1


def nsamples(t, dt) -> int:
    """`round(Int, ustrip(NoUnits, t/dt))`"""
    return round_half_even(_frac(t) / _frac(dt))


def window_index(t, t_first, dt) -> int:
    """0-based sample index of time t: `round(Int,(t-first_x)/step_x)` (tailstats.jl:17-18)."""
    return round_half_even((_frac(t) - _frac(t_first)) / _frac(dt))


def trap_samples(avg, gap, dt, avg2=None) -> _abi.Trap:
    avg2 = avg if avg2 is None else avg2
    return _abi.Trap(nsamples(avg, dt), nsamples(gap, dt), nsamples(avg2, dt))


def sg_npoints(wl, dt) -> int:
    """SavitzkyGolayFilter window: round(wl/dt), made odd (DESIGN.md assumption A2)."""
    n = nsamples(wl, dt)
    return n + 1 if n % 2 == 0 else n


def cuspzac_lowered(rt, ft, tau, length, beta, dt) -> _abi.CuspZac:
    return _abi.CuspZac(float(rt) / float(dt), nsamples(ft, dt), nsamples(length, dt),
                        float(tau) / float(dt), float(beta))


class WindowError(ValueError):
    """A window lies outside the trace — the reference's @assert (tailstats.jl:23-25)."""


def _check_window(name, a, b, n):
    if not (0 <= a <= b <= n - 1):
        raise WindowError(f"{name} window [{a},{b}] outside trace of {n} samples")


DEFAULT_T0_FLT_PARS = (40.0 * ns, 100.0 * ns, 2000.0 * ns)  # src/dsp_routines.jl:9


def lower_icpc(config: DSPConfig, tau: float, pars_filter: dict, L: int, t_first: float, dt: float,
               presum_rate: int = None, windowed: bool = False) -> _abi.IcpcParams:
    """Lower (DSPConfig, tau, pars_filter) + sampling info of the traces to ldsp_icpc_params.

    Follows the parameter unpacking of reference src/dsp_icpc.jl:64-99.  `presum_rate` (dsp_icpc_compressed, :293-350,
    for the presummed traces): upper rail scaled by the rate (:339), one Savitzky-Golay window `sg_wl * rate / 2`
    (:441) — the only SG filter that routine applies to the presummed traces."""
    kw = config.kwargs_pars
    p = _abi.IcpcParams()
    p.L, p.t_first, p.dt, p.unit_per_us = int(L), float(t_first), float(dt), UNIT_PER_US
    bit_depth = int(kw.fc_bit_depth)
    p.sat_low, p.sat_high = 0.0, float(2 ** bit_depth - bit_depth)  # dsp_icpc.jl:94
    if presum_rate is not None:
        p.sat_high *= float(presum_rate)                             # dsp_icpc.jl:339
    p.bl_from = window_index(config.bl_window.left, t_first, dt)
    p.bl_until = window_index(config.bl_window.right, t_first, dt)
    p.tail_from = window_index(config.tail_window.left, t_first, dt)
    p.tail_until = window_index(config.tail_window.right, t_first, dt)
    if windowed:
        p.bl_from, p.bl_until, p.tail_from, p.tail_until = 0, min(7, L - 1), 0, min(7, L - 1)
    _check_window("bl_window", p.bl_from, p.bl_until, L)
    _check_window("tail_window", p.tail_from, p.tail_until, L)
    p.pz_c = float(dt) / float(tau)
    t0p = tuple(kw.t0_flt_pars)
    p.t0_trap = trap_samples(t0p[0], t0p[1], dt, t0p[2])
    p.t0_mintot = max(1, nsamples(kw.t0_mintot, dt))
    p.t0_threshold = float(config.t0_threshold)
    p.t0inv_trap = trap_samples(DEFAULT_T0_FLT_PARS[0], DEFAULT_T0_FLT_PARS[1], dt, DEFAULT_T0_FLT_PARS[2])
    p.tx_mintot = max(1, nsamples(kw.tx_mintot, dt))
    p.int_est = _abi.Dni(nsamples(kw.int_interpolation_length, dt), int(kw.int_interpolation_order))
    p.qdrift_d1, p.qdrift_d2 = config.qdrift_int_length.first, config.qdrift_int_length.last
    p.lq_d1, p.lq_d2 = config.lq_int_length.first, config.lq_int_length.last
    for i, (rt, ft) in enumerate(((10 * us, 4 * us), (5 * us, 3 * us), (3 * us, 1 * us))):
        p.trap_fixed[i] = trap_samples(rt, ft, dt)
    trap_rt, trap_ft = get_fltpars(pars_filter, "trap", config)
    cusp_rt, cusp_ft = get_fltpars(pars_filter, "cusp", config)
    zac_rt, zac_ft = get_fltpars(pars_filter, "zac", config)
    sg_wl = get_fltpars(pars_filter, "sg", config)
    p.trap_opt = trap_samples(trap_rt, trap_ft, dt)
    p.trap_pickoff = float(trap_rt) + float(trap_ft) / 2
    p.sig_est = _abi.Dni(nsamples(kw.sig_interpolation_length, dt), int(kw.sig_interpolation_order))
    tau_off = 10000000.0 * us  # dsp_icpc.jl:98-99 "switch off CR filter"
    p.cusp = cuspzac_lowered(cusp_rt, cusp_ft, tau_off, config.flt_length_cusp, float(config.flt_length_cusp) / float(dt), dt)
    p.zac = cuspzac_lowered(zac_rt, zac_ft, tau_off, config.flt_length_zac, float(config.flt_length_zac) / float(dt), dt)
    p.cusp_pickoff = float(config.flt_length_cusp) / 2
    p.zac_pickoff = float(config.flt_length_zac) / 2
    sg_windows = (sg_wl, 60 * ns, 100 * ns) if presum_rate is None else (float(sg_wl) * presum_rate / 2,) * 3
    for i, wl in enumerate(sg_windows):
        p.sg_npts[i] = sg_npoints(wl, dt)
    p.sg_degree = int(config.sg_flt_degree)
    if presum_rate is not None:
        # the scaled window can be shorter than degree + 1 points (50 ns at 16 ns: 3 points, degree 3 in the reference's
        # own test, test/test_dsp_icpc.jl:164-170): the fit is then the interpolating polynomial.  Both uses of this
        # filter (pile-up threshold in sigmas, t50_current at half maximum) do not depend on its scale.
        p.sg_degree = min(p.sg_degree, p.sg_npts[0] - 1)
        # qdrift / lq are taken from the windowed traces: the integral estimator only has to be well-formed here
        p.int_est = _abi.Dni(max(p.int_est.npts, p.int_est.degree + 1), p.int_est.degree)
    p.cur_left, p.cur_right = config.current_window.left, config.current_window.right
    p.intrace_nsigma = float(config.inTraceCut_std_threshold)
    p.intrace_mintot = max(1, nsamples(kw.intrace_mintot, dt))
    p.bl_left, p.bl_right = config.bl_window.left, config.bl_window.right
    if windowed:
        tiny = trap_samples(dt, 0.0, dt)
        for i in range(3):
            if p.trap_fixed[i].flen > L:
                p.trap_fixed[i] = tiny
        if p.trap_opt.flen + p.sig_est.npts > L:
            p.trap_opt = tiny
        p.cusp = p.zac = _abi.CuspZac(1.0, 0, 5, 1.0e9, 5.0)
        p.bl_left, p.bl_right = 0.0, float(t_first) + (p.sg_npts[0] - 1 + 8) * float(dt)   # 9 samples of the SG axis
    validate_icpc(p)
    return p


def validate_icpc(p: _abi.IcpcParams) -> None:
    """Host-side checks standing in for the reference's @assert / ArgumentError
    paths; raises instead of launching."""
    L = p.L
    if not (64 <= L <= _abi.LDSP_MAX_L):
        raise ValueError(f"trace length {L} outside [64, {_abi.LDSP_MAX_L}]")
    for name, tr in (("t0", p.t0_trap), ("t0inv", p.t0inv_trap), ("trap_opt", p.trap_opt),
                     *[(f"trap_fixed[{i}]", p.trap_fixed[i]) for i in range(3)]):
        if tr.navg < 1 or tr.navg2 < 1 or tr.ngap < 0 or tr.flen > L:
            raise WindowError(f"{name} trapezoid {tr} does not fit a trace of {L} samples")
    for name, cz in (("cusp", p.cusp), ("zac", p.zac)):
        if not (3 <= cz.length <= L) or cz.flat < 0 or cz.flat >= cz.length or cz.sigma <= 0:
            raise WindowError(f"{name} filter (length {cz.length}, flat {cz.flat}) does not fit")
    for name, e in (("int_est", p.int_est), ("sig_est", p.sig_est)):
        if not (e.degree < e.npts <= _abi.LDSP_MAX_EST_PTS) or e.degree > _abi.LDSP_MAX_EST_DEG or e.degree < 0:
            raise ValueError(f"{name}: PolynomialDNI({e.degree}, {e.npts} pts) unsupported")
    for i in range(3):
        n = p.sg_npts[i]
        if not (p.sg_degree < n <= _abi.LDSP_MAX_SG_PTS) or n % 2 == 0:
            raise ValueError(f"SavitzkyGolay window of {n} points (degree {p.sg_degree}) unsupported")
        tf = p.t_first + (n - 1) * p.dt
        a, b = window_index(p.cur_left, tf, p.dt), window_index(p.cur_right, tf, p.dt)
        _check_window("current_window (SG axis)", a, b, L - n + 1)
    _check_window("current_window", window_index(p.cur_left, p.t_first, p.dt),
                  window_index(p.cur_right, p.t_first, p.dt), L)
    n0 = p.sg_npts[0]
    tf = p.t_first + (n0 - 1) * p.dt
    _check_window("bl_window (SG axis)", window_index(p.bl_left + tf, tf, p.dt),
                  window_index(p.bl_right, tf, p.dt), L - n0 + 1)


def lower_sipm(config: dict, pars_optimization: dict, L: int, t_first: float, dt: float) -> _abi.SipmParams:
    """Lower the PropDict config of dsp_sipm (reference src/dsp_sipm.jl:49-78)."""
    config = PropDict(config)
    pars_optimization = PropDict(pars_optimization)
    sg, tr = config.filters.sg, config.filters.trap
    p = _abi.SipmParams()
    p.L, p.t_first, p.dt, p.unit_per_us = int(L), float(t_first), float(dt), UNIT_PER_US
    a, b = config.t0_hpge_window[0], config.t0_hpge_window[-1]
    # TruncateFilter(a..b): the samples whose time lies in the closed interval
    fa = (_frac(a) - _frac(t_first)) / _frac(dt)
    fb = (_frac(b) - _frac(t_first)) / _frac(dt)
    p.trunc_from, p.trunc_until = max(0, math.ceil(fa)), min(L - 1, math.floor(fb))
    _check_window("t0_hpge_window", p.trunc_from, p.trunc_until, L)
    p.sg_npts = sg_npoints(pars_optimization.sg.wl, dt)
    p.sg_degree = int(config.sg_flt_degree)
    if not (p.sg_degree < p.sg_npts <= _abi.LDSP_MAX_SG_PTS):
        raise ValueError(f"SavitzkyGolay window of {p.sg_npts} points unsupported")
    p.sg_mintot = max(1, nsamples(sg.min_tot_intersect, dt))
    p.sg_maxtot = max(1, nsamples(sg.max_tot_intersect, dt))
    p.sg_min_thr, p.sg_max_thr, p.sg_nsigma = sg.min_threshold, sg.max_threshold, sg["n_σ_threshold"]
    p.sg_min_dc_thr, p.sg_max_dc_thr, p.sg_nsigma_dc = sg.min_dc_threshold, sg.max_dc_threshold, sg["n_σ_dc_threshold"]
    p.pz_c = float(dt) / float(tr.pz_tau)
    p.trap = trap_samples(tr.rt, tr.ft, dt)
    p.trap_mintot = max(1, nsamples(tr.min_tot_intersect, dt))
    p.trap_maxtot = max(1, nsamples(tr.max_tot_intersect, dt))
    p.trap_min_thr, p.trap_max_thr, p.trap_nsigma = tr.min_threshold, tr.max_threshold, tr["n_σ_threshold"]
    p.trap_min_dc_thr, p.trap_max_dc_thr, p.trap_nsigma_dc = tr.min_dc_threshold, tr.max_dc_threshold, tr["n_σ_dc_threshold"]
    ng = L - p.sg_npts + 1
    if p.trap.navg < 1 or p.trap.ngap < 0 or p.trap.flen > ng or ng < 8:
        raise WindowError("dsp_sipm trapezoid / SG window does not fit the trace")
    return p


# ---------------------------------------------------------------------------
# The only complete configs in the reference: test/test_dsp_icpc.jl:50-161 and
# test/test_dsp_sipm.jl:38-68.  Used as the default config of bench and tests.

def reference_test_icpc_config() -> DSPConfig:
    grid = {"rt": {"start": 1.0 * us, "stop": 16.0 * us, "step": 0.5 * us},
            "ft": {"start": 1.0 * us, "stop": 4.0 * us, "step": 0.2 * us}}
    return DSPConfig.from_propdict({
        "enc_pickoff_trap": 40.0 * us, "enc_pickoff_zac": 41.0 * us, "enc_pickoff_cusp": 41.0 * us,
        "bl_window": {"min": 0.0 * us, "max": 39.0 * us},
        "tail_window": {"min": 70.0 * us, "max": 110.0 * us},
        "current_window": {"min": 43.0 * us, "max": 62.0 * us},
        "auxbl1_window": {"min": 0.0 * us, "max": 20.0 * us},
        "auxbl2_window": {"min": 20.0 * us, "max": 39.0 * us},
        "auxpz1_window": {"min": 70.0 * us, "max": 90.0 * us},
        "auxpz2_window": {"min": 90.0 * us, "max": 110.0 * us},
        "flt_length_cusp": 38.0 * us, "flt_length_zac": 38.0 * us,
        "t0_threshold": 4.0, "inTraceCut_std_threshold": 5, "sg_flt_degree": 3,
        "qdrift_int_length": (2.5 * us, 0.1 * us, 5.0 * us),
        "lq_int_length": (2.5 * us, 0.1 * us, 5.0 * us),
        "e_grid_trap": grid, "e_grid_zac": grid, "e_grid_cusp": grid,
        "a_grid_wl_sg": {"start": 30.0 * ns, "stop": 350.0 * ns, "step": 32.0 * ns},
        "flt_defaults": {"sg": 100.0 * ns,
                         "trap": {"rt": 5.0 * us, "ft": 2.5 * us},
                         "zac": {"rt": 5.0 * us, "ft": 2.5 * us},
                         "cusp": {"rt": 5.0 * us, "ft": 2.5 * us}},
        "kwargs_pars": {"fc_bit_depth": 16,
                        "t0_flt_pars": [40.0 * ns, 100.0 * ns, 2000.0 * ns],
                        "t0_mintot": 1500.0 * ns, "tx_mintot": 32.0 * ns, "intrace_mintot": 100.0 * ns,
                        "int_interpolation_order": 3, "int_interpolation_length": 100.0 * ns,
                        "sig_interpolation_order": 3, "sig_interpolation_length": 700.0 * ns},
    })


def plumbing_icpc_config_4096() -> DSPConfig:
    """BASELINE config 1 (1 k traces x 4096 samples): the default windows reach 110 us, so a
    4096-sample trace must be sampled at 32 ns.  At 32 ns the fixed SG(60 ns) and the
    100 ns estimator window are 3 points, which cannot carry a cubic: this plumbing config lowers
    sg_flt_degree and int_interpolation_order to 2 (the reference would throw otherwise)."""
    cfg = reference_test_icpc_config()
    cfg.sg_flt_degree = 2
    cfg.kwargs_pars = PropDict(dict(cfg.kwargs_pars, int_interpolation_order=2))
    return cfg


def reference_test_sipm_config() -> PropDict:
    return PropDict({
        "t0_hpge_window": [47.0 * us, 53.0 * us], "sg_flt_degree": 3,
        "filters": {
            "sg": {"n_σ_threshold": 3.0, "min_threshold": -1.0, "max_threshold": 1.0,
                   "n_σ_dc_threshold": 5.0, "min_dc_threshold": -4.0, "max_dc_threshold": 4.0,
                   "min_tot_intersect": 70.0 * ns, "max_tot_intersect": 150.0 * ns},
            "trap": {"rt": 100.0 * ns, "ft": 50.0 * ns, "pz_tau": 3.0 * us,
                     "n_σ_threshold": 3.5, "min_threshold": -1.5, "max_threshold": 1.5,
                     "n_σ_dc_threshold": 5.0, "min_dc_threshold": -3.0, "max_dc_threshold": 3.0,
                     "min_tot_intersect": 48.0 * ns, "max_tot_intersect": 250.0 * ns},
        },
    })
