import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384   # (config 2 at 1 M traces: prof_small.py 1000000 pz_only=1)
opts = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
pz_only = int(opts.pop("pz_only", 0))     # only the config-2 sub-chain (e.g. 524288 traces for its bandwidth counters)
L = 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context()
for k, v in opts.items():
    ctx.set_option(k, int(v))
for _ in range(3):
    if not pz_only:
        ldsp.icpc_run(wf, p, ctx)
    ldsp.icpc_pz_trap_run(wf, p, ctx)
torch.cuda.synchronize()
