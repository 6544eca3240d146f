"""BASELINE config 2, secondary variant: the e_10410 path through the filter-functor entry points (each stage
materialises its output in HBM) vs the fused pz_trap kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 8192
w = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(n, L, device="cuda"), 0.0, 16.0)
def chain():
    st = ldsp.signalstats(w, 0.0, 39000.0)
    x = ldsp.shift_waveform(w, -st["mean"])
    y = ldsp.InvCRFilter(500 * ldsp.us)(x)
    f = ldsp.TrapezoidalChargeFilter(10 * ldsp.us, 4 * ldsp.us)(y)
    return f.signal.max(dim=1).values
chain(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record(); r = chain(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
bytes_alg = n * (4 * L + 4 * (L - 1499))   # functor-materialised variant, 59 540 B per trace
print(f"functor chain (signalstats, shift, InvCR, Trap, max): {best:.2f} ms for {n} traces -> {n/best*1e3/1e6:.1f} Mwf/s, "
      f"{bytes_alg/best*1e3/1e12:.2f} TB/s of the variant's algorithmic bytes ({bytes_alg/best*1e3/8e12*100:.0f}% of 8 TB/s)")
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
ctx = ldsp.default_context(); ctx.enable_timing(True)
ldsp.icpc_pz_trap_run(w.signal, p, ctx); torch.cuda.synchronize()
ms = min((ldsp.icpc_pz_trap_run(w.signal, p, ctx), ctx.last_kernel_ms())[1] for _ in range(3))
print(f"fused pz_trap kernel: {ms:.3f} ms -> {n/ms*1e3/1e6:.1f} Mwf/s; agreement of the two e_10410: max rel diff "
      f"{float(((r - ldsp.icpc_pz_trap_run(w.signal, p, ctx)[1]).abs() / r.abs()).max()):.2e}")
