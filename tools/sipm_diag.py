"""Diagnostic: the two dsp_sipm cases of the GPU suite that disagree with the oracle on counter-based inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import fuzz_cases
orc.build()
np.set_printoptions(precision=6, linewidth=200)
# 1. quantised / degenerate traces
n, L = 24, 16384
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(n, L, device="cuda", seed=77)
wf[:8] = torch.round(wf[:8] * 4) / 4; wf[8:12] = torch.round(wf[8:12]); wf[12] = 0.5; wf[13] = 100.0 * wf[13]
ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=8)
ctx = ldsp.default_context()
for generic in (0, 1):
    ctx.set_option("sipm_generic", generic)
    sc, trig = ldsp.sipm_run(wf, p, ctx)
    torch.cuda.synchronize()
    for c in ("threshold", "threshold_DC", "threshold_trap", "threshold_DC_trap"):
        i = ldsp._abi.SIPM_SCALAR_COLS.index(c)
        print("generic", generic, c, "row 13: gpu %.8f oracle %.8f" % (float(sc[i][13]), ora[c][13]))
ctx.set_option("sipm_generic", 0)
print("dc window", p.sg_min_dc_thr, p.sg_max_dc_thr, p.trap_min_dc_thr, p.trap_max_dc_thr)
# the integrated SG signal of row 13 in float64 and float32, and the valid sets of the DC MAD
x = wf[13].cpu().numpy().astype(np.float64)
# 2. fuzz configuration 4
it = 4
Lf, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(1, it)
print(descr)
p2 = ldsp.lower_sipm(cfg, pf, Lf, 0.0, 16.0)
w2 = fuzz_cases.sipm_traces(192, Lf, it, noise, mean_pulses)
sc, trig = ldsp.sipm_run(w2, p2)
torch.cuda.synchronize()
o2 = orc.dsp_sipm(w2.cpu().numpy(), p2, nthreads=16)
for g in ldsp._abi.SIPM_TRIG_GROUPS:
    cg, co = trig[g]["count"].cpu().numpy(), o2[g]["count"]
    same = cg == co
    for f in ("x", "x_high", "x_tot", "max"):
        a, b = trig[g][f].cpu().numpy().astype(np.float64), o2[g][f]
        d = np.abs(a - b); d[np.isnan(a) & np.isnan(b)] = 0
        rows = np.where(same & (np.nan_to_num(d, nan=1e9).max(axis=1) > (0.01 if f != "max" else 1e-3)))[0]
        for r in rows[:4]:
            k = int(co[r])
            print(g, f, "row", r, "count", k, "gpu", a[r, :k], "oracle", b[r, :k])
    i = ldsp._abi.SIPM_SCALAR_COLS.index({"trig": "threshold", "trig_DC": "threshold_DC", "trig_trap": "threshold_trap", "trig_DC_trap": "threshold_DC_trap"}[g])
    bad = np.where(cg != co)[0]
    print(g, "count mismatches", bad[:8], "threshold gpu/oracle of the first rows above:", [(float(sc[i][r]), o2[ldsp._abi.SIPM_SCALAR_COLS[i]][r]) for r in bad[:3]])
