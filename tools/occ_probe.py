"""Throughput of icpc_lean3_kernel against resident workgroups per CU (option dbg_lds_pad inflates the LDS request).
L = 8192 (512 threads: 1 or 2 workgroups = 2 / 4 waves per SIMD) and L = 4096 at 32 ns (256 threads: 1..4 workgroups =
1..4 waves per SIMD, same VGPR count).  usage: python tools/occ_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp

ctx = ldsp.Context(0); ctx.enable_timing(True)
for L, dt, n, base in ((8192, 16.0, 65536, 80640),):
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, dt)
    wf = ldsp.synth.hpge_batch(n, L, device="cuda")
    out = torch.empty((n, 48), dtype=torch.float32, device="cuda")
    for generic in (0,):
      for pad in (0, 10000, 60000):   # three / two / one workgroup(s) per CU
        ctx.set_option("dbg_lds_pad", pad)
        ts = []
        for _ in range(5):
            ldsp.icpc_run(wf, p, ctx, out=out); ts.append(ctx.last_kernel_ms())
        t = min(ts[1:])
        print(f"L={L} kernel={ctx.last_kernel_name()} lds pad={pad}: {t:.3f} ms  {n / t * 1e-3:.2f} M wf/s  ({n * L / t * 1e-6:.1f} Gsamples/s)", flush=True)
    ctx.set_option("dbg_lds_pad", 0)
