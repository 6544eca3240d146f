"""Which kernel runs dsp_icpc at a set of trace lengths, its rate, and its table against the generic kernel's (debug / timing aid).
usage: python tools/lean_ragged_check.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for L in (8192, 8000, 7300, 6000, 6250, 3000, 1000):
    dt = 16.0
    cfg = ldsp.reference_test_icpc_config()
    try:
        p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, dt)
    except Exception as e:
        print(L, "rejected on the host:", str(e)[:80]); continue
    wf = ldsp.synth.hpge_batch(n, 8192, device="cuda")[:, :L].contiguous()
    ctx = ldsp.Context(0); ctx.enable_timing(True)
    out = {}
    for generic in (0, 1):
        ctx.set_option("icpc_generic", generic)
        tab = ldsp.icpc_run(wf, p, ctx)
        ms = []
        for _ in range(3):
            tab = ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
        out[generic] = (ctx.last_kernel_name(), min(ms), {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()})
    (k0, m0, t0), (k1, m1, t1) = out[0], out[1]
    worst = max((float(np.nanmax(np.abs(t0[c].astype(np.float64) - t1[c].astype(np.float64)) / (1e-3 + np.abs(t1[c].astype(np.float64))))), c) for c in ("blmean", "e_max", "e_10410", "e_trap", "e_cusp", "e_zac", "t50", "t0", "tail_tau", "qdrift", "a_sg"))
    print(f"L={L}: {k0} {m0:.3f} ms = {n / m0 / 1e3:.2f} M wf/s | {k1} {m1:.3f} ms = {n / m1 / 1e3:.2f} M wf/s | worst relative difference {worst[0]:.2e} ({worst[1]})")
