import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
pf = {"cusp": {"rt": 4 * ldsp.us, "ft": 2 * ldsp.us}}
p2 = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, pf, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context(); ctx.enable_timing(True)
def t(fn):
    fn(); torch.cuda.synchronize()
    return min((fn(), ctx.last_kernel_ms())[1] for _ in range(5))
ctx.set_option("dbg_stop", 6); k1 = t(lambda: ldsp.icpc_run(wf, p, ctx))
ctx.set_option("dbg_stop", 0); full = t(lambda: ldsp.icpc_run(wf, p, ctx)); full2 = t(lambda: ldsp.icpc_run(wf, p2, ctx))
pz = t(lambda: ldsp.icpc_pz_trap_run(wf, p, ctx))
ctx.set_option("two_kernel", 1); two = t(lambda: ldsp.icpc_run(wf, p, ctx)); ctx.set_option("two_kernel", 0)
print(f"n={n}: two-kernel path {two:.3f} ms")
print(f"n={n}: phases 0-4 {k1:.3f} ms | full (cusp==zac geometry) {full:.3f} ms -> cz {full-k1:.3f} | full (cusp!=zac) {full2:.3f} ms -> cz x2 {full2-k1:.3f} | pz_trap {pz:.3f} ms")
print(f"  full: {n/full*1e3/1e6:.2f} Mwf/s = {n*32960/full*1e3/8e12*100:.2f}% of 8 TB/s ; pz_trap {n/pz*1e3/1e6:.1f} Mwf/s = {n*32776/pz*1e3/8e12*100:.1f}%")
