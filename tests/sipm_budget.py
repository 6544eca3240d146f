"""What float32 storage of a signal allows a crossing position to differ by, per trigger.

The reference converts SiPM waveforms to Float64 before anything else (src/dsp_sipm.jl:87-88): its integrated trace I, pole-zero
corrected trace and trapezoid output are Float64 arrays, and so are the oracle's.  The HIP kernel keeps ONE copy of the current signal
in LDS and one in registers, both float32 (the sums that build it run with double-precision offsets; what is rounded is the STORED
sample).  A crossing position is  t0 + dt * (p - 1 + (th - s[p-1]) / (s[p] - s[p-1])):  a stored sample that is off by half a
float32 ulp of its level moves it by  dt * ulp(level) / (2 |slope|)  — nothing on a steep edge, tenths of a nanosecond on the shallow
crossings of a slow discharge.  The threshold itself (n_sigma x a MAD of float32 samples whose Savitzky-Golay arithmetic is float32:
its error integrates to ~1e-6 in I, tools/sipm_budget_diag.py) differs from the oracle's by a few 1e-6, which moves every crossing of
the trace by that difference over the slope.  `position_budget` evaluates both terms for ONE trigger from the float64 restatement of
the chain, so that a test accepts exactly what these two causes explain and nothing else (tests/test_sipm_gpu.py, tests/fuzz_cases.py)."""
import numpy as np

ULPS = 4.0   # half an ulp for each of the two samples, the threshold (a median of stored samples times a factor) and the float32 partial
             # sums inside a wave-row: 4 ulps of the level in all
FLOOR_NS = 0.01


def signals64(x, p, orc):
    """g (Savitzky-Golay derivative), I (integrator), the pole-zero corrected trace and the trapezoid output of ONE trace in float64,
    by the oracle's functors (the statements of oracle/ldsp_oracle.c: sipm_one)."""
    h = orc.sg_coeffs(p.sg_npts, p.sg_degree, 1)
    g = orc.fir(np.asarray(x, dtype=np.float64), h)
    i_ = orc.integrator(g)
    pz = orc.invcr(i_, p.pz_c)
    t = orc.trap(pz, p.trap.navg, p.trap.ngap, p.trap.navg2)
    return g, i_, pz, t


def group_signal(group, sig, p):
    """(signal the group's crossings are found on, time of its first sample)"""
    g, i_, pz, t = sig
    tg = p.t_first + (p.sg_npts - 1) * p.dt
    if group == "trig":
        return g, tg
    if group in ("trig_DC", "trig_DC_trap"):
        return -i_, tg
    return t, tg + (p.trap.navg + p.trap.ngap + p.trap.navg2 - 1) * p.dt


def group_level(group, sig, p):
    """(the float32 signal whose rounding reaches the group's samples, how many of its samples enter one of them, the factor between its
    level and the level of the sums the kernel forms of it).  The trapezoid output is a difference of two box sums of the pole-zero
    corrected trace P = I + c cumsum(I), formed in float32 (sipm_s4.inc: box_pass): its own level says nothing about the rounding it
    carries — with a short pz_tau, P is orders of magnitude above the trapezoid's output, and a box sum of n samples rounds at n times
    P's level."""
    g, i_, pz, t = sig
    if group == "trig_trap":
        return pz, p.trap.navg + p.trap.ngap + p.trap.navg2, float(max(p.trap.navg, p.trap.navg2))
    return (g if group == "trig" else i_), 1, 1.0


def position_budget(s, t0, dt, x_ns, lev=None, span=1, gain=1.0, dth=0.0):
    """Bound (ns) on the difference of a crossing found at x_ns on the float64 signal s when the samples it is made of (lev, `span` of
    them per sample of s, first one at the same index) are stored as float32 and the two thresholds differ by dth."""
    u = (x_ns - t0) / dt
    if not np.isfinite(u):
        return FLOOR_NS
    k = int(np.clip(np.ceil(u), 1, len(s) - 1))
    slope = abs(s[k] - s[k - 1])
    if slope == 0.0:
        return np.inf
    lev = s if lev is None else lev
    # (the level of the WHOLE trace: I and P are running sums, and the rounding of their largest partial sums — float32 inside a
    # wave-row — is carried forward to every later sample)
    level = gain * float(np.max(np.abs(lev)))
    ulp = float(np.spacing(np.float32(level)))
    return max(FLOOR_NS, dt * (ULPS * ulp + abs(dth)) / slope)


def trigger_budgets(group, x_row, p, orc, ora_row, dth=0.0):
    """Budgets for the fields x, x_high, x_tot of every trigger of one trace (ora_row: the oracle's dict of this group's fields for the row;
    dth: n_sigma x the difference of the kernel's and the oracle's MAD threshold of this trace — the threshold column is held to its own
    tolerance, and what is left of it moves every crossing of the trace by dth / slope)."""
    sig = signals64(x_row, p, orc)
    s, t0 = group_signal(group, sig, p)
    lev, span, gain = group_level(group, sig, p)
    n = len(ora_row["x"])
    bx = np.array([position_budget(s, t0, p.dt, ora_row["x"][j], lev, span, gain, dth) for j in range(n)])
    bh = np.array([position_budget(s, t0, p.dt, ora_row["x_high"][j], lev, span, gain, dth) for j in range(n)])
    return {"x": bx, "x_high": bh, "x_tot": bx + bh}


THR_COL = {"trig": ("threshold", "sg_nsigma"), "trig_DC": ("threshold_DC", "sg_nsigma_dc"), "trig_trap": ("threshold_trap", "trap_nsigma"),
           "trig_DC_trap": ("threshold_DC_trap", "trap_nsigma_dc")}


def count_flip_explained(group, x_row, p, orc, thr_gpu, thr_ora):
    """A trace whose trigger COUNT differs: is there a sample of the float64 signal that sits between the two thresholds (n_sigma x the
    GPU's and the oracle's MAD threshold) or within the float32 resolution of them?  Then one crossing more or less is what float32
    storage gives; otherwise the difference is a defect."""
    sig = signals64(x_row, p, orc)
    s, _ = group_signal(group, sig, p)
    lev, span, gain = group_level(group, sig, p)
    ns = float(getattr(p, THR_COL[group][1]))
    lo, hi = sorted((ns * float(thr_gpu), ns * float(thr_ora)))
    tol = ULPS * float(np.spacing(np.float32(max(gain * float(np.max(np.abs(lev))), abs(hi)))))
    return bool(np.any((s >= lo - tol) & (s <= hi + tol)))


def compare_triggers(trig, ora, wf_host, p, orc, sc_gpu=None, scalar_cols=None, fields=("x", "x_high", "x_tot", "max")):
    """Per trigger group: rows whose count differs and is / is not explained by a sample at the threshold, rows with a position beyond
    its float32-storage budget, rows with a maximum beyond tolerance.  trig: the kernel's groups (torch tensors or arrays)."""
    res = {}
    for g in THR_COL:
        cg = np.asarray(trig[g]["count"].cpu() if hasattr(trig[g]["count"], "cpu") else trig[g]["count"])
        co = ora[g]["count"]
        flips, defects, posbad, maxbad = [], [], [], []
        for r in np.nonzero(cg != co)[0]:
            thr_o = ora[THR_COL[g][0]][r]
            thr_g = thr_o if sc_gpu is None else float(sc_gpu[scalar_cols.index(THR_COL[g][0])][r])
            (flips if orc is not None and count_flip_explained(g, wf_host[r], p, orc, thr_g, thr_o) else defects).append(int(r))
        arr = {f: np.asarray(trig[g][f].cpu() if hasattr(trig[g][f], "cpu") else trig[g][f]).astype(np.float64) for f in fields}
        for r in np.nonzero((cg == co) & (co > 0))[0]:
            c = int(co[r])
            row = {f: ora[g][f][r][:c] for f in ("x", "x_high", "x_tot")}
            d = {f: np.abs(arr[f][r][:c] - ora[g][f][r][:c]) for f in fields}
            if any((d[f] > FLOOR_NS).any() for f in ("x", "x_high", "x_tot") if f in d):
                if orc is None:   # (no oracle at hand, e.g. the frozen vectors: the flat 0.01 ns)
                    posbad.append(int(r))
                    continue
                thr_o = ora[THR_COL[g][0]][r]
                thr_g = thr_o if sc_gpu is None else float(sc_gpu[scalar_cols.index(THR_COL[g][0])][r])
                b = trigger_budgets(g, wf_host[r], p, orc, row, dth=float(getattr(p, THR_COL[g][1])) * (thr_g - thr_o))
                if any((d[f] > b[f]).any() for f in ("x", "x_high", "x_tot") if f in d):
                    posbad.append(int(r))
            if "max" in d and (d["max"] > 1e-3 + 1e-4 * np.abs(ora[g]["max"][r][:c])).any():
                maxbad.append(int(r))
        res[g] = {"flips": flips, "count_defects": defects, "positions": posbad, "maxima": maxbad}
    return res


def mad_gap_tolerance(v, lo, hi, ranks=2):
    """thresholdstats_mad (src/thresholdstats.jl:61-71) = 1.4826 * median(|y - median(y)|) over the samples lo <= y <= hi.  ONE sample
    whose float32 value falls on the other side of a window bound than its float64 value changes the count by one and moves each of
    the two medians to a neighbouring order statistic: the threshold may then differ by 1.4826 x (the spacing of the order statistics
    around the median of y + the spacing around the median of the deviations) — returned here, over +-`ranks` ranks, for the float64
    signal v.  On a well-populated window this is ~1e-5; on a trace with a few hundred samples left inside the window it is ~1e-2."""
    y = np.sort(v[(v >= lo) & (v <= hi)])
    if len(y) < 2 * ranks + 2:
        return np.inf
    k = (len(y) - 1) // 2
    med = 0.5 * (y[k] + y[len(y) // 2])
    gap_y = y[min(k + ranks, len(y) - 1)] - y[max(k - ranks, 0)]
    d = np.sort(np.abs(y - med))
    gap_d = d[min(k + ranks, len(d) - 1)] - d[max(k - ranks, 0)]
    return 1.4826 * (gap_y + gap_d)


def threshold_window(col, p):
    """(which signal, lower bound, upper bound) of the MAD window behind a threshold column"""
    return {"threshold": ("trig", p.sg_min_thr, p.sg_max_thr), "threshold_DC": ("trig_DC", p.sg_min_dc_thr, p.sg_max_dc_thr),
            "threshold_trap": ("trig_trap", p.trap_min_thr, p.trap_max_thr), "threshold_DC_trap": ("trig_DC_trap", p.trap_min_dc_thr, p.trap_max_dc_thr)}[col]
