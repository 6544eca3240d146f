"""tools/sipm_cols_ab.py L [LIB ...] — the scalar columns of dsp_sipm for the first rows of a seeded batch of L-sample traces from each library build
(and from the oracle), side by side."""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, R)
    import numpy as np, torch
    import legenddsp_jl_amd as ldsp
    L = int(sys.argv[2])
    p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
    wf = ldsp.synth.sipm_batch(32, L, device="cuda", seed=9)
    sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
    out = {c: sc[i].cpu().numpy()[:3].astype(float).tolist() for i, c in enumerate(ldsp._abi.SIPM_SCALAR_COLS)}
    for g in ("trig_DC", "trig_trap"):
        out[g + ".x[0]"] = trig[g]["x"][0][:int(trig[g]["count"][0])].cpu().numpy().astype(float).tolist()[:6]
    if len(sys.argv) > 3:
        from oracle import oracle as orc
        orc.build()
        ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=8)
        out = {c: np.asarray(ora[c])[:3].astype(float).tolist() for c in ldsp._abi.SIPM_SCALAR_COLS}
        for g in ("trig_DC", "trig_trap"):
            out[g + ".x[0]"] = np.asarray(ora[g]["x"][0][:int(ora[g]["count"][0])]).astype(float).tolist()[:6]
    print(json.dumps(out)); sys.exit(0)
L = sys.argv[1]
res = {}
for lib in sys.argv[2:] + ["oracle"]:
    env = dict(os.environ, LDSP_ALLOW_STALE="1")
    if lib != "oracle": env["LDSP_HIP_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, __file__, "--child", L] + (["o"] if lib == "oracle" else []), env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line: print(lib, r.stdout[-500:], r.stderr[-1500:]); continue
    res[os.path.basename(lib)] = json.loads(line[-1])
for c in next(iter(res.values())):
    print(f"{c:22s}", "  ".join(f"{k}: " + " ".join(f"{v:+.6e}" for v in res[k][c]) for k in res))
