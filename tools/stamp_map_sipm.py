"""Timeline of k_sipm_s4 from in-kernel s_memtime stamps (diagnostic build: tools/dev_build_sipm.sh stamps -DLDSP_STAMPS; run with
LDSP_HIP_LIB=build/dev/libldsp_stamps.so).  Per interval between two consecutive stamps of a wave: mean cycles, the mean wait of a
wave for the slowest one of its workgroup at the closing stamp, share of the lifetime.  usage: python tools/stamp_map_sipm.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, NW, SLOTS, BLOCKS = 16384, 8, 64, 1024
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(n, L, device="cuda")
ctx = ldsp.Context(0)
buf = torch.zeros((BLOCKS, 16, SLOTS), dtype=torch.int64, device="cuda")
ldsp.sipm_run(wf, p, ctx)
ctx.set_option("dbg_stamps", buf.data_ptr())
ldsp.sipm_run(wf, p, ctx)
torch.cuda.synchronize()
ctx.set_option("dbg_stamps", 0)
s = buf.cpu().numpy()[:, :NW, :].astype(np.float64)
nst = int((s[0, 0] > 0).sum())
same = ((s > 0).sum(axis=2) == nst).all(axis=1)
s = s[same][:, :, :nst]
print(f"{same.sum()} of {BLOCKS} stamped workgroups with {nst} stamps per wave; cycles (s_memtime ticks, 100 MHz) per wave")
med = ["statistics", "m1 hist issued", "m1 hist complete", "m1 located", "m1 candidates", "m1 selected", "m2 hist issued", "m2 hist complete", "m2 located", "m2 candidates", "m2 selected"]
names = ["load + extremes", "SG"] + ["MAD(SG) " + x for x in med] + ["stage end", "mask + trig (SG)", "integrate + stats"]
names += ["MAD(DC sg) " + x for x in med] + ["MAD(DC trap) " + x for x in med] + ["DC triggers (both)", "InvCR + trap"] + ["MAD(trap) " + x for x in med] + ["trig (trap) + end"]
life = (s[:, :, -1].max(axis=1) - s[:, :, 0].min(axis=1)).mean()
for j in range(1, nst):
    d = (s[:, :, j] - s[:, :, j - 1]).mean()
    spread = (s[:, :, j].max(axis=1, keepdims=True) - s[:, :, j]).mean()
    nm = names[j - 1] if j - 1 < len(names) else f"stamp {j}"
    print(f"{j:3d} {nm:36s} {d:8.0f} {spread:8.0f} {100 * d / life:6.1f}%")
print(f"workgroup lifetime {life:8.0f}")
