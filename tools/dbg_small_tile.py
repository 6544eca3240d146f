"""debug: the small-tile case (1800 samples, uint16, separate CUSP / ZAC) of tests/test_icpc_gpu.py run several times — are the tables
of repeated launches identical, and which rows / columns differ from the oracle by a NaN?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, dataclasses
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import parity
from test_icpc_gpu import _scaled_config
length, dt, u16, sep = int(sys.argv[1]) if len(sys.argv) > 1 else 1800, 16.0, True, True
us = ldsp.us
cfg, sc = _scaled_config(length, dt, degree2=False)
pf = {"cusp": {"rt": 4.0 * us * sc, "ft": 1.5 * us * sc}, "zac": {"rt": 5.5 * us * sc, "ft": 2.0 * us * sc}} if sep else {}
pf["trap"] = {"rt": 5.0 * us * sc, "ft": 2.5 * us * sc}
p = ldsp.lower_icpc(cfg, 500 * us, pf, length, 0.0, dt)
n = 128
wf = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=71)
idx = (torch.arange(length, device="cuda", dtype=torch.float32) * (8192.0 / length)).long().clamp(max=8191)
wf = wf[:, idx].contiguous().round().clamp(0, 65535)
host = wf.cpu().numpy()
ora = orc.dsp_icpc(host, p, nthreads=16)
ctx = ldsp.default_context()
tabs = []
for rep in range(6):
    out = torch.full((n, 48), float("nan"), device="cuda")
    t = ldsp.icpc_run(wf.to(torch.uint16) if (rep % 2 == 0) else wf, p, ctx, out=out)
    torch.cuda.synchronize()
    tabs.append(t.clone())
print("kernel", ctx.last_kernel_name())
for rep in range(1, 6):
    d = (tabs[rep].view(torch.int32) != tabs[0].view(torch.int32))
    print("rep", rep, "differs from rep 0 in", int(d.sum()), "cells; columns", sorted(set(ldsp._abi.ICPC_COLS[j] for j in d.nonzero()[:, 1].tolist())))
g = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tabs[0]).items()}
for c in ldsp._abi.ICPC_COLS:
    a, b = np.asarray(g[c], dtype=np.float64), np.asarray(ora[c], dtype=np.float64)
    m = np.isnan(a) != np.isnan(b)
    if m.any():
        print(c, "NaN mismatch rows", np.nonzero(m)[0][:10], "gpu", a[m][:5], "oracle", b[m][:5])
    bad, err = parity.bad_mask(c, g, ora, host, p, orc)
    if bad.any():
        print(c, "bad rows", np.nonzero(bad)[0][:10], "gpu", a[bad][:5], "oracle", b[bad][:5])
for c in ("t0", "t80", "qdrift", "lq", "e_max"):
    print(c, g[c][:4], ora[c][:4])
