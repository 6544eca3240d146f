import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import parity, fuzz_cases
orc.build()
for seed, it in ((7, 1), (7, 20), (7, 10), (9, 13), (5, 24)):
    L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(seed, it, wide=True)
    p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    wf = fuzz_cases.icpc_traces(256, L, it, noise)
    for generic in (0, 1):
        ctx = ldsp.default_context(); ctx.set_option("icpc_generic", generic)
        tab = ldsp.icpc_run(wf, p); torch.cuda.synchronize()
        ctx.set_option("icpc_generic", 0)
        gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
        host = wf.cpu().numpy()
        ora = orc.dsp_icpc(host, p, nthreads=16, strict=False)
        bad, _ = parity.bad_mask("t0_inv", gpu, ora, host, p, orc)
        r = np.nonzero(bad)[0]
        print(seed, it, descr[:40], "generic" if generic else "lean", "rows", r.tolist(), "gpu", gpu["t0_inv"][r].tolist(), "oracle", np.asarray(ora["t0_inv"])[r].tolist(), "t0", gpu["t0"][r].tolist(), flush=True)
