"""One launch of the fused dsp_icpc kernel per profiling stop (dbg_stop), in a fixed order, for
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -- python3 tools/prof_phases.py`:
the differences between consecutive dispatches are the dynamic instruction counts of each phase
(tools/prof_phases_report.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context()
for stop in (1, 7, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 0):
    ctx.set_option("dbg_stop", stop)
    ldsp.icpc_run(wf, p, ctx)
    torch.cuda.synchronize()
