import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
orc.build()
n, L = 256, 16384
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(n, L, device="cuda")
sc, trig = ldsp.sipm_run(wf, p)
torch.cuda.synchronize()
ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16)
for g in ldsp._abi.SIPM_TRIG_GROUPS:
    cg, co = trig[g]["count"].cpu().numpy(), ora[g]["count"]
    same = cg == co
    for f in ("x", "x_high", "x_tot", "max"):
        a, b = trig[g][f].cpu().numpy().astype(np.float64), ora[g][f]
        d = np.abs(a - b)[same]
        d = d[np.isfinite(d)]
        print(g, f, trig[g][f].dtype, "n", d.size, "median %.3g  p99 %.3g  max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()) if d.size else "")
