import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(n, L, device="cuda")
ctx = ldsp.default_context(); ctx.enable_timing(True)
sc, trig = ldsp.sipm_run(wf, p, ctx); torch.cuda.synchronize()
ms = min((ldsp.sipm_run(wf, p, ctx), ctx.last_kernel_ms())[1] for _ in range(3))
elems = sum(int(trig[g]["count"].clamp(max=64).sum()) for g in trig) * 4
b = n * (4 * L + 80) + 4 * elems
print(f"dsp_sipm n={n} L={L}: {ms:.3f} ms -> {n/ms*1e3/1e6:.2f} Mwf/s, {b/ms*1e3/1e12:.3f} TB/s = {b/ms*1e3/8e12*100:.1f}% of 8 TB/s; ragged elements/trace {elems/n:.1f}")
prev = 0
for stop, name in ((1, "load + extremestats x2"), (2, "SG filter"), (3, "MAD threshold (SG)"), (4, "mask + IntersectMaximum (SG)"), (5, "integrator + signalstats x2"),
                   (6, "2 x (MAD + mask + IntersectMaximum) on -I"), (7, "InvCR + trapezoid"), (0, "MAD + IntersectMaximum (trap)")):
    ctx.set_option("dbg_stop", stop)
    ldsp.sipm_run(wf, p, ctx); torch.cuda.synchronize()
    t = min((ldsp.sipm_run(wf, p, ctx), ctx.last_kernel_ms())[1] for _ in range(2))
    print(f"  after {name}: {t:.2f} ms (+{t-prev:.2f})"); prev = t
