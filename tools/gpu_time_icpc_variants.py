"""Kernel rate of dsp_icpc for the parameter sets beside the headline's, with the kernel each one ran and a hash of its table (so that
two library builds can be compared bit for bit): trace lengths that do not fill the tile, lengths that are no multiple of four samples,
optimised Savitzky-Golay windows of 11 ... 27 taps, CUSP and ZAC optimised separately, uint16 input.
Usage (GPU box): [LDSP_HIP_LIB=build/dev/libldsp_X.so] python tools/gpu_time_icpc_variants.py [n_traces]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
us, ns = ldsp.us, ldsp.ns
sep = {"cusp": {"rt": 4.0 * us, "ft": 1.5 * us}, "zac": {"rt": 5.5 * us, "ft": 2.0 * us}}
cases = [("headline", 8192, {}, False), ("L=8000", 8000, {}, False), ("L=8190", 8190, {}, False), ("L=7001", 7001, {}, False),
         ("SG 180 ns (11 taps)", 8192, {"sg": {"wl": 180.0 * ns}}, False), ("SG 240 ns (15 taps)", 8192, {"sg": {"wl": 240.0 * ns}}, False),
         ("SG 300 ns (19 taps)", 8192, {"sg": {"wl": 300.0 * ns}}, False), ("SG 350 ns (23 taps)", 8192, {"sg": {"wl": 350.0 * ns}}, False),
         ("SG 400 ns (25 taps)", 8192, {"sg": {"wl": 400.0 * ns}}, False), ("SG 430 ns (27 taps)", 8192, {"sg": {"wl": 430.0 * ns}}, False),
         ("separate CUSP / ZAC", 8192, sep, False), ("separate CUSP / ZAC, L=8000", 8000, sep, False),
         ("separate CUSP / ZAC, SG 300 ns", 8192, dict(sep, sg={"wl": 300.0 * ns}), False), ("uint16 input", 8192, {}, True),
         ("uint16 input, separate CUSP / ZAC, L=8190", 8190, sep, True)]
ctx = ldsp.Context(0)
ctx.enable_timing(True)
full = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=3)
for name, L, pf, u16 in cases:
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, pf, L, 0.0, 16.0)
    wf = full[:, :L].contiguous()
    if u16:
        wf = wf.round().clamp(0, 65535).to(torch.uint16)
    out = torch.empty((n, len(ldsp._abi.ICPC_COLS)), dtype=torch.float32, device="cuda")
    ms = []
    for it in range(7):
        ldsp.icpc_run(wf, p, ctx, out=out)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms())
    ms = sorted(ms[2:])
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{name:44s} {ctx.last_kernel_name():26s} min {ms[0]:7.3f} ms  median {ms[len(ms) // 2]:7.3f} ms  {n / ms[len(ms) // 2] / 1e3:6.2f} M waveforms/s  table {h}", flush=True)
    del wf, out
