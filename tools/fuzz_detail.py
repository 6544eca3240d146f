"""Per-trace detail of the columns that disagree in one case of the randomised dsp_icpc sweep (tests/fuzz_cases.py).
Usage (GPU box): python tools/fuzz_detail.py CASE [seed] [option=value ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import parity, fuzz_cases
orc.build()
it = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 256
L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(seed, it)
p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
wf = fuzz_cases.icpc_traces(n, L, it, noise)
ctx = ldsp.default_context()
for kv in sys.argv[3:]:      # context options, e.g. icpc_generic=1 cusp_direct=1
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tab = ldsp.icpc_run(wf, p); torch.cuda.synchronize()
print(f"[{it}] {descr} kernel={ctx.last_kernel_name()}")
gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
host = wf.cpu().numpy()
ora = orc.dsp_icpc(host, p, nthreads=16, strict=False)
lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
print(f"worst bad fraction {worst:.4f}")
for c in ldsp._abi.ICPC_COLS:
    bad, err = parity.bad_mask(c, gpu, ora, host, p, orc)
    a, b = np.asarray(gpu[c], dtype=np.float64), np.asarray(ora[c], dtype=np.float64)
    if bad.any():
        idx = np.nonzero(bad)[0]
        print(f"{c}: {len(idx)} bad rows")
        for i in idx[:12]:
            print(f"    row {i:3d}: gpu {a[i]:.9g}  oracle {b[i]:.9g}  diff {a[i]-b[i]:.3g}   [t0 {ora['t0'][i]:.4f} e_max {ora['e_max'][i]:.6g} blsigma {ora['blsigma'][i]:.4g} tailslope {ora['tailslope'][i]:.4g} t0_inv {ora['t0_inv'][i]:.4g}]")
