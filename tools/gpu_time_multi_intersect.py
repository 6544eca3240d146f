"""MultiIntersect(0.01:0.01:0.90, mintot) on n x 8192 baseline-subtracted HPGe traces (reference src/multi_intersect.jl):
the wave-parallel search (one wave per threshold) against the reference's one-lane walk (option multi_serial).
usage (GPU box): python tools/gpu_time_multi_intersect.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L, DT = 8192, 16.0
wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=5)
wf -= wf[:, :2000].mean(dim=1, keepdim=True)
w = ldsp.ArrayOfRDWaveforms(wf, 0.0, DT)
f = ldsp.MultiIntersect(threshold_ratios=tuple(np.arange(0.01, 0.905, 0.01)), mintot=4 * DT, n=1, d=1, sampling_rate=1)
ctx = ldsp.default_context()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for serial in (0, 1):
    ctx.set_option("multi_serial", serial)
    out = f(w); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); out = f(w); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    res[serial] = out
    b = n * (4 * L + 4 * 90)
    print(f"MultiIntersect K=90 mintot=4 samples, {n} x {L} ({'one-lane walk' if serial else 'wave per threshold'}): {best:.2f} ms -> "
          f"{n / best * 1e3 / 1e6:.2f} M waveforms/s, {b / best * 1e3 / 1e12:.2f} TB/s of 4L + 4K bytes per trace ({b / best * 1e3 / 8e12 * 100:.1f} % of 8 TB/s)")
ctx.set_option("multi_serial", 0)
print("bit-identical results:", bool(torch.equal(res[0], res[1])))
