"""Column-wise comparison of the HIP dsp_icpc table with the CPU oracle.

Tolerances (float32 compute vs float64 oracle), per column class:
  stat / energy columns : |a-b| <= atol_c + RTOL*|b|, RTOL = 2e-5
  crossing times (us)   : |a-b| <= 5e-4 us (0.03 sample of 16 ns)
  argmax times (ns)     : equal, or the two maxima tie within the energy tolerance
  integer columns       : exact
A threshold decision |y - thr| below float32 resolution can legitimately flip;
such traces show up as isolated outliers and are reported, and the tests bound
their fraction (FLIP_FRAC).
"""
import numpy as np

from legenddsp_jl_amd import _abi

RTOL = 2e-5
FLIP_FRAC = 0.01

TIME_US = ["t0", "t10", "t50", "t80", "t90", "t99", "t50_current", "t0_inv"]
TIME_MAX = ["t_trap_max", "t_cusp_max", "t_zac_max"]
INT_COLS = _abi.ICPC_I32_COLS
# absolute floors per column (units of the column); energies scale with amplitude
ATOL = {
    "blmean": 2e-3, "blsigma": 2e-4, "blslope": 1e-8, "bloffset": 2e-3,
    "tailmean": 0.05, "tailsigma": 5e-3, "tailslope": 2e-7, "tailoffset": 0.05,
    "tail_tau": 50.0, "tail_mean": 2e-6, "tail_sigma": 2e-6,
    "e_max": 2e-3, "e_min": 2e-3,
    "e_10410": 0.05, "e_535": 0.05, "e_313": 0.05, "e_10410_inv": 0.05, "e_313_inv": 0.1,
    "e_trap": 0.05, "e_cusp": 0.1, "e_zac": 0.1, "e_trap_max": 0.05, "e_cusp_max": 0.1, "e_zac_max": 0.1,
    # qdrift / lq: the integrator is taken relative to the first window point (csrc/qdrift.hpp); was 40 with float32 prefix sums
    "qdrift": 3.0, "lq": 3.0, "a_sg": 5e-3, "a_60": 5e-3, "a_100": 5e-3, "a_raw": 5e-3,
    "drift_time": 0.6, "inTrace_intersect": 0.6,
}


def compare(gpu: dict, ora: dict, verbose=False):
    """gpu / ora: dict column -> numpy array.  Returns (report_lines, worst_bad_fraction)."""
    lines, worst = [], 0.0
    n = len(next(iter(ora.values())))
    for c in _abi.ICPC_COLS:
        a = np.asarray(gpu[c], dtype=np.float64)
        b = np.asarray(ora[c], dtype=np.float64)
        both_nan = np.isnan(a) & np.isnan(b)
        if c in INT_COLS:
            bad = (a != b) & ~both_nan
            err = np.abs(a - b)
        elif c in TIME_US:
            err = np.abs(a - b)
            bad = ~(err <= 5e-4) & ~both_nan
        elif c in TIME_MAX:
            err = np.abs(a - b)
            bad = ~(err <= 1e-3) & ~both_nan
            # near-tie: accept when the corresponding maxima agree
            mcol = c.replace("t_", "e_")
            tie = np.abs(np.asarray(gpu[mcol], dtype=np.float64) - ora[mcol]) <= ATOL[mcol] + RTOL * np.abs(ora[mcol])
            bad &= ~tie
        else:
            err = np.abs(a - b)
            bad = ~(err <= ATOL[c] + RTOL * np.abs(b)) & ~both_nan
        frac = bad.sum() / max(n, 1)
        worst = max(worst, frac)
        e = err[~both_nan & np.isfinite(err)]
        scale = np.nanmax(np.abs(b)) if np.isfinite(b).any() else 0
        lines.append(f"{c:20s} max|err|={e.max() if e.size else 0:11.4g}  ref scale={scale:11.4g}  bad={int(bad.sum())}/{n}")
    return lines, worst
