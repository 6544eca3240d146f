"""tools/sipm_thr_ab.py LIB_A LIB_B seed it — the scalar table and trigger counts of one randomised dsp_sipm configuration (tests/fuzz_cases.py) from two
library builds, compared bit for bit (each library in a process of its own)."""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 4 and sys.argv[1] == "--child":
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import numpy as np, torch, hashlib
    import legenddsp_jl_amd as ldsp
    import fuzz_cases
    seed, it = int(sys.argv[2]), int(sys.argv[3])
    L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(seed, it)
    p = ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0)
    wf = fuzz_cases.sipm_traces(192, L, it, noise, mean_pulses)
    sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
    out = {"scalars": hashlib.sha1(sc.cpu().numpy().tobytes()).hexdigest()}
    for g in ldsp._abi.SIPM_TRIG_GROUPS:
        out[g] = hashlib.sha1(trig[g]["count"].cpu().numpy().tobytes()).hexdigest()[:10]
    thr = {c: sc[ldsp._abi.SIPM_SCALAR_COLS.index(c)].cpu().numpy().view(np.int32).tolist() for c in ("threshold", "threshold_DC", "threshold_DC_trap", "threshold_trap")}
    print(json.dumps({"hash": out, "thr": thr}))
    sys.exit(0)
a, b, seed, it = sys.argv[1:5]
res = []
for lib in (a, b):
    env = dict(os.environ, LDSP_HIP_LIB=os.path.abspath(lib), LDSP_ALLOW_STALE="1")
    r = subprocess.run([sys.executable, __file__, "--child", seed, it], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(r.stdout, r.stderr); sys.exit(1)
    res.append(json.loads(line[-1]))
print("hashes equal:", res[0]["hash"] == res[1]["hash"], res[0]["hash"], res[1]["hash"])
for c in res[0]["thr"]:
    d = [i for i, (x, y) in enumerate(zip(res[0]["thr"][c], res[1]["thr"][c])) if x != y]
    print(c, "rows that differ:", d[:10])
