import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context(); ctx.enable_timing(True)
prev = 0
for stop in (1, 7, 2, 3, 4, 5, 6):
    ctx.set_option("dbg_stop", stop)
    ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize()
    ms = min((ldsp.icpc_run(wf, p, ctx), ctx.last_kernel_ms())[1] for _ in range(3))
    print(f"stop after phase {stop}: {ms:.3f} ms (+{ms-prev:.3f})  {ms/n*1e3:.3f} us/wf"); prev = ms

# stops inside the CUSP/ZAC stage (fused: same launch; kernel 1's phases run in full before it)
base = prev
for stop, name in ((11, "y ready"), (12, "A0 Dp + A1 flat top/u"), (13, "d + B causal"), (14, "C anti-causal"), (15, "A2 parabola"), (0, "finish")):
    ctx.set_option("dbg_stop", stop)
    ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize()
    ms = min((ldsp.icpc_run(wf, p, ctx), ctx.last_kernel_ms())[1] for _ in range(3))
    print(f"cz stop {stop:2d} ({name}): {ms:.3f} ms (+{ms-prev:.3f})"); prev = ms
