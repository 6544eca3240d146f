import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 16384
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(n, L, device="cuda")
ctx = ldsp.default_context()
if len(sys.argv) > 2: ctx.set_option("dbg_stop", int(sys.argv[2]))
for _ in range(3):
    ldsp.sipm_run(wf, p, ctx)
torch.cuda.synchronize()
