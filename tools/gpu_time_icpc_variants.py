"""dsp_icpc throughput for parameter sets the lean kernel does and does not take: shared CUSP/ZAC geometry (the reference's
test configuration) vs separately optimised rise / flat-top times per filter (pars_filter), L = 8192.  usage: [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L, us = 8192, ldsp.us
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context(); ctx.enable_timing(True)
cases = {"test configuration (CUSP = ZAC geometry)": {},
         "separate rise times: cusp rt 4.0 / zac rt 5.5 us": {"cusp": {"rt": 4.0 * us, "ft": 1.5 * us}, "zac": {"rt": 5.5 * us, "ft": 1.5 * us}},
         "separate flat tops: cusp ft 1.0 / zac ft 2.0 us": {"cusp": {"rt": 5.0 * us, "ft": 1.0 * us}, "zac": {"rt": 5.0 * us, "ft": 2.0 * us}}}
for name, pf in cases.items():
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, pf, L, 0.0, 16.0)
    ldsp.icpc_run(wf, p, ctx); torch.cuda.synchronize()
    ms = min((ldsp.icpc_run(wf, p, ctx), ctx.last_kernel_ms())[1] for _ in range(3))
    print(f"{name:55s} {ctx.last_kernel_name():24s} {ms:.3f} ms -> {n / ms * 1e3 / 1e6:.2f} M waveforms/s")
# uint16 ADC counts: converted in the kernel's load vs a separate cast pass before the float32 kernel
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, {}, L, 0.0, 16.0)
w16 = wf.round().clamp(0, 65535).to(torch.uint16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
t_a = timed(lambda: ldsp.icpc_run(w16, p, ctx))
t_b = timed(lambda: ldsp.icpc_run(w16.to(torch.float32), p, ctx))
print(f"uint16 input, converted while loading: {t_a:.3f} ms -> {n / t_a * 1e3 / 1e6:.2f} M waveforms/s;  cast pass + float32 kernel: {t_b:.3f} ms -> {n / t_b * 1e3 / 1e6:.2f} M waveforms/s")
# BASELINE config 2 (pole-zero + trapezoid) on float32 and on uint16 input, at a batch that does not fit the caches
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
wf2 = ldsp.synth.hpge_batch(65536, L, device="cuda").round().clamp(0, 65535).repeat(n2 // 65536, 1)
w2_16 = wf2.to(torch.uint16)
out2 = torch.empty((2, n2), dtype=torch.float32, device="cuda")
for name, x, bpt in (("float32", wf2, 4 * L + 8), ("uint16", w2_16, 2 * L + 8)):
    ldsp.icpc_pz_trap_run(x, p, ctx, out=out2); torch.cuda.synchronize()
    ms = min((ldsp.icpc_pz_trap_run(x, p, ctx, out=out2), ctx.last_kernel_ms())[1] for _ in range(5))
    print(f"pole-zero + trapezoid, {name} input, {n2} traces: {ctx.last_kernel_name()} {ms:.3f} ms -> {n2 / ms * 1e3 / 1e6:.1f} M waveforms/s, "
          f"{n2 * bpt / ms * 1e3 / 1e12:.2f} TB/s = {n2 * bpt / ms * 1e3 / 8e12 * 100:.0f} % of 8 TB/s ({bpt} B per trace)")
