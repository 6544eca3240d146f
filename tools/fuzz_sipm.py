"""Randomised parity sweep of the fused dsp_sipm kernels against the oracle: SG window, trapezoid, pole-zero constant,
threshold windows, n-sigma factors, time-over-threshold limits, trace length, noise, pulse density, discharges (negative
pulses), ADC-like quantisation.  Usage (GPU box): python tools/fuzz_sipm.py [n_configs] [seed]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
orc.build()
nconf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
n = 192
names = ldsp._abi.SIPM_SCALAR_COLS
tot_bad = 0
for it in range(nconf):
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
    try:
        p = ldsp.lower_sipm(cfg, {"sg": {"wl": wl}}, L, 0.0, 16.0)
    except Exception as e:
        print(f"[{it}] rejected on the host: {type(e).__name__}: {e}"); continue
    noise = float(rng.choice([0.1, 0.3, 1.0]))
    wf = ldsp.synth.sipm_batch(n, L, device="cuda", seed=500 + it, noise=noise, mean_pulses=float(rng.choice([0.5, 3.0, 8.0])))
    wf[:16] -= ldsp.synth.sipm_batch(16, L, device="cuda", seed=900 + it, noise=0.0, mean_pulses=2.0) * 3.0     # discharges
    wf[16:32] = torch.round(wf[16:32] * 8) / 8
    sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
    ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16)
    msgs = []
    for i, c in enumerate(names):
        a, b = sc[i].cpu().numpy().astype(np.float64), ora[c]
        tol = 2e-3 + 1e-4 * np.abs(b)
        if c in ("blslope", "wfslope"): tol = 1e-7 + 1e-4 * np.abs(b)
        if c.startswith("t_"): tol = 1e-3
        bad = ~(np.abs(a - b) <= tol) & ~(np.isnan(a) & np.isnan(b))
        if bad.any(): msgs.append(f"{c}: {int(bad.sum())}/{n} max|err| {np.nanmax(np.abs(a - b)[bad]):.3g}")
    for g in ldsp._abi.SIPM_TRIG_GROUPS:
        cg, co = trig[g]["count"].cpu().numpy(), ora[g]["count"]
        diff = int((cg != co).sum())
        same = cg == co
        xa, xb = trig[g]["x"].cpu().numpy().astype(np.float64), ora[g]["x"]
        okx = (np.abs(xa - xb) <= 0.05) | (np.isnan(xa) & np.isnan(xb))
        xbad = int((~okx[same]).any(axis=1).sum())
        if diff or xbad: msgs.append(f"{g}: count differs on {diff}, positions on {xbad} of {n} (mean count {co.mean():.1f})")
    tot_bad += len(msgs)
    print(f"[{it}] L={L} wl={wl:.0f} trap=({tr['rt']:.0f},{tr['ft']:.0f}) pz_tau={tr['pz_tau']:.0f} noise={noise}: " + ("; ".join(msgs) if msgs else "all within tolerance"))
print("entries with any disagreement:", tot_bad)
