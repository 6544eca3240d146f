"""Randomised dsp_icpc / dsp_sipm configurations and trace batches shared by tests/test_fuzz_gpu.py (fixed seeds) and the
open-ended sweeps tools/fuzz_icpc.py / tools/fuzz_sipm.py.  Case `it` of seed `seed` draws from default_rng([seed, it]),
so a case is reproducible on its own."""
import copy
import dataclasses

import numpy as np
import torch

import legenddsp_jl_amd as ldsp


def icpc_case(seed, it, wide=False):
    """-> (L, dt, cfg, tau, pars_filter, noise, description).  Filter parameters, tau, window placement, trace length.
    wide (round 4; the cases without it are unchanged): trace lengths that are no multiple of four samples and optimised
    Savitzky-Golay windows from the whole of the reference's scan grid (up to 350 ns: 23 taps), one step beyond it (400 ns: 25
    taps, the lean kernel's limit) and past it (430 ns: the generic kernel)."""
    rng = np.random.default_rng([seed, it])
    us = ldsp.us
    L = int(rng.choice([8192, 8192, 6000, 7300, 16384]))
    dt = 16.0
    pf = {"trap": {"rt": float(rng.uniform(2, 12)) * us, "ft": float(rng.uniform(0.5, 4)) * us},
          "sg": {"wl": float(rng.choice([80, 100, 132, 180, 200])) * ldsp.ns}}
    if wide:
        rw = np.random.default_rng([seed, it, 77])
        L = int(rw.choice([8190, 8191, 7301, 6002, 4099, 5003, 8189])) if wide != 4 else int(rw.choice([8188, 8184, 7300, 6004, 4100, 5004, 8180]))   # (wide = 4: the same sweep with rows that ARE 16-byte aligned, as a control)
        pf["sg"]["wl"] = float(rw.choice([62, 126, 222, 254, 318, 350, 400, 430])) * ldsp.ns
    crt, cft = float(rng.uniform(2, 8)) * us, float(rng.uniform(0.5, 3)) * us
    pf["cusp"] = {"rt": crt, "ft": cft}
    zac_same = bool(rng.random() < 0.6)
    pf["zac"] = dict(pf["cusp"]) if zac_same else {"rt": float(rng.uniform(2, 8)) * us, "ft": float(rng.uniform(0.5, 3)) * us}
    tau = float(rng.uniform(150, 900)) * us
    sc = L * dt / 131072.0
    fl_cusp = 38.0 * us * sc * rng.uniform(0.6, 1.0)
    fl_zac = fl_cusp if zac_same else 38.0 * us * sc * rng.uniform(0.6, 1.0)
    cfg = dataclasses.replace(ldsp.reference_test_icpc_config(),
                              bl_window=ldsp.ClosedInterval(0.0, 39.0 * us * sc * rng.uniform(0.7, 1.0)),
                              tail_window=ldsp.ClosedInterval(70.0 * us * sc, 110.0 * us * sc * rng.uniform(0.9, 1.0)),
                              current_window=ldsp.ClosedInterval(43.0 * us * sc, 62.0 * us * sc),
                              flt_length_cusp=fl_cusp, flt_length_zac=fl_zac)
    noise = float(rng.choice([0.0, 1.0, 3.0, 10.0]))
    descr = (f"L={L} tau={tau / us:.0f}us trap=({pf['trap']['rt'] / us:.1f},{pf['trap']['ft'] / us:.1f}) cusp=({crt / us:.1f},{cft / us:.1f}) "
             f"zac_same={zac_same} sg={pf['sg']['wl']:.0f}ns noise={noise}")
    return L, dt, cfg, tau, pf, noise, descr


def icpc_traces(n, L, it, noise):
    """HPGe pulses plus pile-up on the tail (rows 0-7), flat-topped (16-19) and rail-saturated (20-21) traces."""
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=1000 + it, noise=noise)
    wf[:8] = wf[:8] + torch.roll(wf[8:16] - wf[8:16, :1], 900, dims=1) * 0.5
    wf[16:20] = wf[16:20].clamp(max=65520.0 * 0.1 + 900)
    wf[20:22] = (wf[20:22] * 8).clamp(min=0.0, max=65520.0)
    return wf


def sipm_case(seed, it):
    """-> (L, cfg, pars_filter, noise, mean_pulses, description)."""
    rng = np.random.default_rng([seed, it])
    L = int(rng.choice([16384, 16384, 8192, 6250, 12000, 4096]))
    cfg = copy.deepcopy(dict(ldsp.reference_test_sipm_config()))
    sg, tr = cfg["filters"]["sg"], cfg["filters"]["trap"]
    wl = float(rng.choice([100, 150, 200, 250]))
    tr["rt"], tr["ft"] = float(rng.choice([48, 100, 160, 200])), float(rng.choice([0, 50, 100]))
    tr["pz_tau"] = float(rng.uniform(500, 6000))
    w = float(rng.uniform(0.6, 2.5)); sg["min_threshold"], sg["max_threshold"] = -w, w
    w = float(rng.uniform(1.0, 3.0)); tr["min_threshold"], tr["max_threshold"] = -w, w
    w = float(rng.uniform(2.0, 6.0)); sg["min_dc_threshold"], sg["max_dc_threshold"] = -w, w
    if rng.random() < 0.5:
        w = float(rng.uniform(2.0, 6.0))
    tr["min_dc_threshold"], tr["max_dc_threshold"] = -w, w
    sg["n_σ_threshold"], tr["n_σ_threshold"] = float(rng.uniform(2.5, 5)), float(rng.uniform(2.5, 5))
    sg["min_tot_intersect"], sg["max_tot_intersect"] = float(rng.choice([32, 70, 100])), float(rng.choice([150, 300]))
    tr["min_tot_intersect"], tr["max_tot_intersect"] = float(rng.choice([32, 48, 100])), float(rng.choice([250, 500]))
    span = L * 16.0
    cfg["t0_hpge_window"] = [0.45 * span, 0.52 * span]
    noise = float(rng.choice([0.1, 0.3, 1.0]))
    mean_pulses = float(rng.choice([0.5, 3.0, 8.0]))
    descr = f"L={L} wl={wl:.0f} trap=({tr['rt']:.0f},{tr['ft']:.0f}) pz_tau={tr['pz_tau']:.0f} noise={noise} pulses={mean_pulses}"
    return L, cfg, {"sg": {"wl": wl}}, noise, mean_pulses, descr


def sipm_traces(n, L, it, noise, mean_pulses):
    """SiPM pulse trains plus discharges (negative pulses, rows 0-15) and ADC-like quantisation (rows 16-31)."""
    wf = ldsp.synth.sipm_batch(n, L, device="cuda", seed=500 + it, noise=noise, mean_pulses=mean_pulses)
    wf[:16] -= ldsp.synth.sipm_batch(16, L, device="cuda", seed=900 + it, noise=0.0, mean_pulses=2.0) * 3.0
    wf[16:32] = torch.round(wf[16:32] * 8) / 8
    return wf


def sipm_compare(sc, trig, ora, n, wf=None, p=None, orc=None):
    """Messages for every dsp_sipm column / trigger group that disagrees with the oracle (empty list = parity).  Trigger positions: ONE
    rule for all four groups — 0.01 ns, or, on a shallow crossing, what half an ulp of the float32-stored samples moves the interpolated
    position by, evaluated per trigger from the float64 restatement (tests/sipm_budget.py); a trigger count may differ only where a sample
    of that restatement lies AT the threshold, and on at most 2 % of the traces."""
    import sipm_budget
    msgs = []
    for i, c in enumerate(ldsp._abi.SIPM_SCALAR_COLS):
        a, b = sc[i].cpu().numpy().astype(np.float64), ora[c]
        tol = 2e-3 + 1e-4 * np.abs(b)
        if c in ("blslope", "wfslope"):
            tol = 1e-7 + 1e-4 * np.abs(b)
        if c.startswith("t_"):
            tol = 1e-3
        bad = ~(np.abs(a - b) <= tol) & ~(np.isnan(a) & np.isnan(b))
        if bad.any():
            msgs.append(f"{c}: {int(bad.sum())}/{n} max|err| {np.nanmax(np.abs(a - b)[bad]):.3g}")
    host = wf.cpu().numpy().astype(np.float32)
    res = sipm_budget.compare_triggers(trig, ora, host, p, orc, sc_gpu=[x.cpu().numpy() for x in sc], scalar_cols=ldsp._abi.SIPM_SCALAR_COLS,
                                       fields=("x", "x_high", "x_tot"))
    for g, r in res.items():
        if r["count_defects"] or r["positions"] or len(r["flips"]) > max(1, n // 50):
            msgs.append(f"{g}: count differs with no sample at the threshold on {r['count_defects'][:6]}, at the threshold on {len(r['flips'])} of {n}, "
                        f"positions beyond the float32-storage budget on {r['positions'][:6]}")
    return msgs
