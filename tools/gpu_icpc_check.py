"""Run the fused dsp_icpc kernel on a synthetic batch and print its agreement with the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
direct = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = 8192
cfg = ldsp.reference_test_icpc_config()
p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context()
ctx.set_option("cusp_direct", direct)
t = time.time(); tab = ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize(); print("gpu first call s", time.time() - t)
t = time.time(); tab = ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize(); print("gpu second call s", time.time() - t)
g = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
t = time.time(); o = orc.dsp_icpc(wf.cpu().numpy(), p, nthreads=16); print("oracle s", time.time() - t)
lines, worst = parity.compare(g, o)
print("\n".join(lines)); print("worst bad fraction", worst)
for c in ["e_10410_inv", "e_313_inv", "a_raw", "t0", "t50", "e_trap", "e_cusp", "e_zac", "qdrift", "lq", "a_sg", "inTrace_intersect", "t50_current", "tail_tau"]:
    print(c, g[c][:4], o[c][:4]); bad = np.argmax(np.abs(g[c] - o[c])); print("   worst idx", bad, g[c][bad], o[c][bad])
pz = ldsp.icpc_pz_trap_run(wf, p, ctx).cpu().numpy()
o2 = orc.icpc_pz_trap(wf.cpu().numpy(), p)
print("pz_trap blmean err", np.abs(pz[0] - o2["blmean"]).max(), "e_10410 rel err", (np.abs(pz[1] - o2["e_10410"]) / o2["e_10410"]).max())
