"""Rate of the fused dsp_icpc kernel on the bench configuration (A/B of library builds through LDSP_HIP_LIB).  usage: python tools/lean_time.py [n] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, 8192, device="cuda")
ctx = ldsp.Context(0); ctx.enable_timing(True)
out = torch.empty((n, 48), dtype=torch.float32, device="cuda")
ms = []
for _ in range(reps + 2):
    ldsp.icpc_run(wf, p, ctx, out=out); torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
ms = sorted(ms[2:])
print(f"{os.environ.get('LDSP_HIP_LIB', 'production')}: {ctx.last_kernel_name()} min {ms[0]:.3f} ms median {ms[len(ms) // 2]:.3f} ms = {n / ms[len(ms) // 2] / 1e3:.3f} M waveforms/s")
