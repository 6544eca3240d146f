"""A/B timing + parity check of dsp_icpc variants in ONE process (interleaved rounds; cdna_hip_programming.md rule 24).
usage: python tools/dev_time.py [n] [opt=val ...]   e.g.  python tools/dev_time.py 65536 icpc_r2=1
Runs the fused kernel with the default options and with the given options alternately (5 rounds), prints min / median
kernel ms of both, and the parity of the variant against the oracle on the first 512 traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp

n = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 65536
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("base:"))
BASE_OPTS = dict(a[5:].split("=") for a in sys.argv[1:] if a.startswith("base:"))   # options set in both runs, e.g. base:icpc_generic=1
L = 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
ctx = ldsp.Context(0); ctx.enable_timing(True)
for k, v in BASE_OPTS.items():
    ctx.set_option(k, int(v))
out = torch.empty((n, 48), dtype=torch.float32, device="cuda")

def setopts(on):
    for k, v in opts.items():
        ctx.set_option(k, int(v) if on else 0)

ts = {False: [], True: []}
for rnd in range(6):
    for on in ((False, True) if opts else (False,)):
        setopts(on)
        ldsp.icpc_run(wf, p, ctx, out=out)
        if rnd:
            ts[on].append(ctx.last_kernel_ms())
for on in ((False, True) if opts else (False,)):
    t = np.array(ts[on])
    print(f"{'variant ' + str(opts) if on else 'default':40s} min {t.min():.3f} ms  median {np.median(t):.3f} ms  -> {n / t.min() * 1e-3:.2f} M waveforms/s")
# parity of both configurations
from oracle import oracle as orc
import parity
m = min(n, 512)
o = orc.dsp_icpc(wf[:m].cpu().numpy(), p, nthreads=16)
tabs = {}
for on in ((False, True) if opts else (False,)):
    setopts(on)
    ldsp.icpc_run(wf, p, ctx, out=out)
    torch.cuda.synchronize()
    tabs[on] = out[:4096].clone()
    g = {k: v[:m].cpu().numpy() for k, v in ldsp.table_columns(out).items()}
    lines, worst = parity.compare(g, o)
    print(f"{'variant' if on else 'default'}: parity vs oracle on {m} traces: worst bad fraction {worst:.4f}")
    if worst > 0:
        print("\n".join(l for l in lines if not l.rstrip().endswith("bad=0/%d" % m)))
    if os.environ.get("DEV_TIME_VERBOSE"):
        print("\n".join(lines))
if opts:
    a, b = tabs[False].cpu().numpy(), tabs[True].cpu().numpy()
    cols = ldsp._abi.ICPC_COLS
    diff = [(c, float(np.nanmax(np.abs(a[:, i].astype(np.float64) - b[:, i]))), int((a[:, i].view(np.int32) != b[:, i].view(np.int32)).sum())) for i, c in enumerate(cols)]
    print("default vs variant, per column (max |diff|, rows whose bits differ): " + ", ".join(f"{c} {d:.3g}/{k}" for c, d, k in diff if k))
