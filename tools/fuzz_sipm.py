"""Open-ended randomised parity sweep of the fused dsp_sipm kernels against the oracle (the cases of tests/fuzz_cases.py;
tests/test_fuzz_gpu.py pins seed 1, cases 0-5).  Usage (GPU box): python tools/fuzz_sipm.py [n_configs] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import fuzz_cases
orc.build()
nconf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 192
tot_bad = 0
for it in range(nconf):
    L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(seed, it)
    try:
        p = ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0)
    except Exception as e:
        print(f"[{it}] {descr}: rejected on the host: {type(e).__name__}: {e}"); continue
    wf = fuzz_cases.sipm_traces(n, L, it, noise, mean_pulses)
    sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
    ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16)
    msgs = fuzz_cases.sipm_compare(sc, trig, ora, n, wf, p, orc)
    tot_bad += len(msgs)
    print(f"[{it}] {descr}: " + ("; ".join(msgs) if msgs else "all within tolerance"), flush=True)
print("entries with any disagreement:", tot_bad)
