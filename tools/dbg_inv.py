import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
L=8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(8, L, device="cuda")
ctx = ldsp.default_context(); ctx.set_option("dbg_stop", 0)
tab = ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize()
g = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
o = orc.dsp_icpc(wf.cpu().numpy(), p)
print("oracle e_10410_inv", o["e_10410_inv"][:4])
print("gpu e_10410_inv   ", g["e_10410_inv"][:4])
print("dbg direct atomic min (drift_time col)", g["drift_time"][:4])
print("dbg slot F0 (t0_inv col)", g["t0_inv"][:4])
x = wf.cpu().numpy().astype(np.float64)
for i in range(2):
    xs = x[i] - o["blmean"][i]
    y = orc.invcr(xs, p.pz_c)
    tr = orc.trap(y, 625, 250)
    print("oracle trap min/max", tr.min(), tr.max(), "argmin", tr.argmin(), len(tr))
