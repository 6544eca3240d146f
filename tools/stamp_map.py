"""Timeline of the fused dsp_icpc kernel from in-kernel s_memtime stamps (diagnostic build: tools/dev_build.sh stamps
-DLDSP_STAMPS; run with LDSP_HIP_LIB=build/dev/libldsp_stamps.so).  Per phase (interval between two stamps): mean cycles
a wave spends in it, the part of it that is waiting for the slowest wave of the workgroup (arrival spread at the stamp
that ends the phase), and the share of the trace's lifetime.  usage: python tools/stamp_map.py [n_traces]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import legenddsp_jl_amd as ldsp

n = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 65536
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)   # context options, e.g. dbg_lds_pad=60000 (one workgroup per CU)
L, NW, SLOTS, BLOCKS = 8192, 8, 32, 2048
params = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.Context(0)
for k, v in opts.items():
    ctx.set_option(k, int(v))
buf = torch.zeros((BLOCKS, 16, SLOTS), dtype=torch.int64, device="cuda")
out = torch.empty((n, 48), dtype=torch.float32, device="cuda")
ldsp.icpc_run(wf, params, ctx, out=out)          # warm-up without stamps
ctx.set_option("dbg_stamps", buf.data_ptr())
ldsp.icpc_run(wf, params, ctx, out=out)
torch.cuda.synchronize()
ctx.set_option("dbg_stamps", 0)
s = buf.cpu().numpy()[:, :NW, :].astype(np.float64)     # [block, wave, slot]
generic = {0: "start", 1: "load + raw sums", 2: "bl reduce + barrier", 3: "blmean, saturation", 4: "shift, tail logs, cumsum scan",
           5: "pz + y -> LDS + barrier", 6: "SG pass", 7: "SG reductions + barrier", 8: "wvf maxima, SG masks", 9: "T scan + T -> LDS + barrier",
           10: "sweep A (7 masks)", 11: "sweep B (4 trapezoids)", 12: "sweep reductions + barrier", 13: "run scans + barrier",
           14: "crossings", 15: "estimators + barrier", 16: "CZ: Dp (+barrier)", 17: "CZ: flat top + ZAC taps", 18: "CZ: d, causal scan + readback",
           19: "CZ: anti-causal scan + readback", 20: "CZ: double cumsum + readback", 21: "CZ: finish", 22: "CZ: collect"}
lean = {0: "start", 1: "load, raw extremes, baseline sums", 2: "per-wave partials + barrier", 3: "blmean, shift, tail sums, cumsum scan + barrier",
        4: "pz offsets (wave 0) + barrier, pole-zero", 5: "y -> LDS, wvf maxima + barrier", 6: "SG pass (S4, packed)", 7: "SG reductions",
        8: "LS pass: sg50 crossing, in-trace mask", 9: "T scan (wave 0) + T -> LDS + barriers", 10: "sweep A (t0 masks)",
        11: "sweep B (4 trapezoids)", 12: "sweep reductions + barrier", 13: "run scans on the masks + barrier",
        14: "threshold confirmation, crossings", 15: "estimators, parabolas (waves 0-3), to the barrier", 16: "barrier, CZ: Dp, d (+barrier)",
        17: "CZ: flat top + ZAC taps (summed by parts)", 18: "CZ: causal scan + readback", 19: "CZ: anti-causal scan + readback",
        20: "CZ: double cumsum + readback", 21: "CZ: maxima, estimator points", 22: "CZ: collect"}
lean3 = {0: "start", 1: "load, raw extremes, baseline sums, prefix scans of x'", 2: "barrier, row tables (wave 0) | blmean", 3: "barrier, y and T -> X, tail logs, candidates",
         4: "barrier, sweep A (S4: t0 masks)", 5: "sweep B (4 trapezoids)", 6: "tail sums, SG pass (registers, DPP halo)", 7: "the round's reductions",
         8: "barrier", 9: "y -> X, SG masks from registers", 10: "barrier", 11: "run scans on the masks, threshold confirmation", 12: "barrier",
         13: "t50_current / pile-up position, finishing lanes, crossings", 14: "estimators, parabolas (waves 0-3)", 15: "yprev, p0, barrier",
         16: "CZ: Dp -> X + barrier",
         17: "CZ: u (ZAC tap chain), flat top, u -> X + 2 barriers", 18: "CZ: double cumsum -> X + 2 barriers, readback", 19: "CZ: d",
         20: "CZ: causal scan + readback", 21: "CZ: anti-causal scan + readback", 22: "CZ: maxima, estimator points", 23: "CZ: collect"}
kn = ctx.last_kernel_name()
names = lean3 if "lean3" in kn else lean if "lean" in kn else generic
print("kernel:", ctx.last_kernel_name())
ids = sorted(names)
valid = (s[:, :, ids] > 0).all(axis=(1, 2))
s = s[valid]
print(f"{valid.sum()} of {BLOCKS} stamped workgroups complete; cycles per wave (mean over waves and workgroups)")
life = (s[:, :, ids[-1]].max(axis=1) - s[:, :, ids[0]].min(axis=1)).mean()
tot = 0.0
print(f"{'phase':42s} {'mean':>8s} {'wait':>8s} {'share':>7s}   arrival of waves 0.. after the first one")
for a, b in zip(ids[:-1], ids[1:]):
    d = (s[:, :, b] - s[:, :, a]).mean()
    # time the average wave then waits for the slowest one of its workgroup at the end of this phase
    spread = (s[:, :, b].max(axis=1, keepdims=True) - s[:, :, b]).mean()
    tot += d
    late = (s[:, :, b] - s[:, :, b].min(axis=1, keepdims=True)).mean(axis=0)      # per wave: arrival after the first wave
    print(f"{names[b]:42s} {d:8.0f} {spread:8.0f} {100 * d / life:6.1f}%   " + " ".join(f"{v:5.0f}" for v in late))
print(f"{'workgroup lifetime (first start -> last end)':42s} {life:8.0f}")
