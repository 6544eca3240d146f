"""Randomised parity sweep of the fused dsp_icpc kernel against the oracle: filter parameters, tau, window placement, trace
length / sampling step, amplitudes, noise, pile-up and saturated traces.  Prints the worst column of every configuration.
Usage (GPU box): python tools/fuzz_icpc.py [n_configs] [seed]"""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import parity
orc.build()
nconf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
n = 256
bad_total = 0
for it in range(nconf):
    L = int(rng.choice([8192, 8192, 6000, 7300, 16384]))
    dt = 16.0
    cfg = ldsp.reference_test_icpc_config()
    us = ldsp.us
    pf = {"trap": {"rt": float(rng.uniform(2, 12)) * us, "ft": float(rng.uniform(0.5, 4)) * us},
          "sg": {"wl": float(rng.choice([80, 100, 132, 180, 200])) * ldsp.ns}}
    crt, cft = float(rng.uniform(2, 8)) * us, float(rng.uniform(0.5, 3)) * us
    pf["cusp"] = {"rt": crt, "ft": cft}
    pf["zac"] = {"rt": crt, "ft": cft} if rng.random() < 0.6 else {"rt": float(rng.uniform(2, 8)) * us, "ft": float(rng.uniform(0.5, 3)) * us}
    tau = float(rng.uniform(150, 900)) * us
    span = L * dt
    sc = span / 131072.0
    cfg = dataclasses.replace(cfg, bl_window=ldsp.ClosedInterval(0.0, 39.0 * us * sc * rng.uniform(0.7, 1.0)),
                              tail_window=ldsp.ClosedInterval(70.0 * us * sc, 110.0 * us * sc * rng.uniform(0.9, 1.0)),
                              current_window=ldsp.ClosedInterval(43.0 * us * sc, 62.0 * us * sc),
                              flt_length_cusp=38.0 * us * sc * rng.uniform(0.6, 1.0), flt_length_zac=38.0 * us * sc * rng.uniform(0.6, 1.0) if pf["zac"] is not pf["cusp"] else None)
    if cfg.flt_length_zac is None or pf["zac"] == pf["cusp"]:
        cfg = dataclasses.replace(cfg, flt_length_zac=cfg.flt_length_cusp)
    try:
        p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    except Exception as e:
        print(f"[{it}] L={L} config rejected on the host: {type(e).__name__}: {e}")
        continue
    noise = float(rng.choice([0.0, 1.0, 3.0, 10.0]))
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=1000 + it, noise=noise)
    wf[:8] = wf[:8] + torch.roll(wf[8:16] - wf[8:16, :1], 900, dims=1) * 0.5                 # pile-up on the tail
    wf[16:20] = wf[16:20].clamp(max=65520.0 * 0.1 + 900)                                     # flat-topped
    wf[20:22] = (wf[20:22] * 8).clamp(min=0.0, max=65520.0)                                  # saturated on the rail
    tab = ldsp.icpc_run(wf, p); torch.cuda.synchronize()
    gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
    ora = orc.dsp_icpc(wf.cpu().numpy(), p, nthreads=16, strict=False)
    lines, worst = parity.compare(gpu, ora)
    bad = [l for l in lines if not l.rstrip().endswith(f"bad=0/{n}")]
    bad_total += len(bad)
    print(f"[{it}] L={L} dt={dt} tau={tau/us:.0f}us trap=({pf['trap']['rt']/us:.1f},{pf['trap']['ft']/us:.1f}) cusp=({crt/us:.1f},{cft/us:.1f}) "
          f"zac_same={pf['zac']==pf['cusp']} sg={pf['sg']['wl']:.0f}ns noise={noise}  worst bad fraction {worst:.4f}")
    for l in bad:
        print("     ", l)
print("columns with any disagreement:", bad_total)
