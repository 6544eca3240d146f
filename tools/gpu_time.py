import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 8192
cfg = ldsp.reference_test_icpc_config()
p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context(); ctx.enable_timing(True)
for name, fn in (("pz_trap", lambda: ldsp.icpc_pz_trap_run(wf, p, ctx)), ("icpc", lambda: ldsp.icpc_run(wf, p, ctx))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        fn(); ts.append(ctx.last_kernel_ms())
    ms = min(ts)
    print(f"{name}: n={n} {ms:.3f} ms  {n/ms*1e3/1e6:.2f} Mwf/s  {n*L*4/ms*1e3/1e12:.3f} TB/s  ({n*L*4/ms*1e3/8e12*100:.1f}% of 8 TB/s)")
