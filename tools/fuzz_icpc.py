"""Open-ended randomised parity sweep of the fused dsp_icpc kernel against the oracle (the cases of tests/fuzz_cases.py;
tests/test_fuzz_gpu.py pins seed 1, cases 0-5).  Prints the worst column of every configuration.
Usage (GPU box): python tools/fuzz_icpc.py [n_configs] [seed] [wide]   (wide: odd trace lengths and long Savitzky-Golay windows, fuzz_cases.icpc_case)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import parity, fuzz_cases
orc.build()
nconf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wide = (4 if sys.argv[3] == "wide4" else sys.argv[3] == "wide") if len(sys.argv) > 3 else False
n = 256
bad_total = 0
for it in range(nconf):
    L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(seed, it, wide=wide)
    try:
        p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    except Exception as e:
        print(f"[{it}] {descr}: config rejected on the host: {type(e).__name__}: {e}")
        continue
    wf = fuzz_cases.icpc_traces(n, L, it, noise)
    tab = ldsp.icpc_run(wf, p); torch.cuda.synchronize()
    gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16, strict=False)
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    bad = [l for l in lines if f"bad=0/{n}" not in l]
    bad_total += len(bad)
    print(f"[{it}] {descr} sg_taps={p.sg_npts[0]} kernel={ldsp.default_context().last_kernel_name()}  worst bad fraction {worst:.4f}", flush=True)
    for l in bad:
        print("     ", l)
print("columns with any disagreement:", bad_total)
