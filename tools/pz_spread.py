"""tools/pz_spread.py — why do warm launches of lean::pz_trap_lean_kernel (BASELINE config 2) range 5.6 .. 6.7 ms in the rocprofv3 trace while every
other kernel is stable to 0.5 %?  Per-launch durations (torch events on the launch stream, read after ONE synchronisation) of 24 launches
(a) back to back, (b) with a host synchronisation after every launch, (c) with a 20 ms idle gap before every launch, (d) back to back right
after a 1 M-trace dsp_icpc launch (what a bench step sequence does), (e) on a second, disjoint input buffer alternating with the first."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n, L = 1_000_000, 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.default_context()
out = torch.empty((n, 48), dtype=torch.float32, device="cuda")

def series(k, gap=0.0, sync_each=False, bufs=None):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        if gap:
            torch.cuda.synchronize(); time.sleep(gap)
        a.record()
        ldsp.icpc_pz_trap_run(bufs[i % len(bufs)] if bufs else wf, p, ctx)
        b.record()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]

def show(name, t):
    t2 = t[2:]
    print(f"{name:64s} min {min(t2):.3f}  max {max(t2):.3f}  mean {sum(t2) / len(t2):.3f} ms   first two {t[0]:.3f} {t[1]:.3f}   all: " + " ".join(f"{x:.2f}" for x in t))

for _ in range(3):
    ldsp.icpc_pz_trap_run(wf, p, ctx)
torch.cuda.synchronize()
show("(a) back to back", series(24))
show("(b) host synchronisation after every launch", series(24, sync_each=True))
show("(c) 20 ms idle before every launch", series(24, gap=0.02))
ldsp.icpc_run(wf, p, ctx, out=out)
show("(d) back to back right after a dsp_icpc launch", series(24))
wf2 = wf.clone()
show("(e) two input buffers alternating (64 GB in flight)", series(24, bufs=[wf, wf2]))
