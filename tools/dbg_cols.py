"""Print rows where the lean and the generic kernel disagree on given columns (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
cols = sys.argv[1:] or ["t50_current"]
n, L = 512, 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.Context(0)
a = {k: v.cpu().numpy() for k, v in ldsp.table_columns(ldsp.icpc_run(wf, p, ctx)).items()}
ctx.set_option("icpc_generic", 1)
b = {k: v.cpu().numpy() for k, v in ldsp.table_columns(ldsp.icpc_run(wf, p, ctx)).items()}
o = orc.dsp_icpc(wf.cpu().numpy(), p, nthreads=16)
for c in cols:
    bad = np.nonzero(~np.isclose(a[c], b[c], rtol=1e-4, atol=1e-3))[0]
    print(c, "rows differing:", len(bad), bad[:20])
    for i in bad[:12]:
        print(f"   row {i}: lean {a[c][i]:.6g} generic {b[c][i]:.6g} oracle {o[c][i]:.6g}   t50 {o['t50'][i]:.4f} e_max {o['e_max'][i]:.1f} a_sg {o['a_sg'][i]:.2f}")
