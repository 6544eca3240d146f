"""tools/sipm_pos_check.py [n_configs seed] — dsp_sipm against the float64 oracle: |difference| of every matched trigger's x / x_high / x_tot (ns) and
maximum, and the relative difference of the four MAD thresholds, on the reference configuration (256 x 16384 traces) and, with arguments, over
the randomised configurations of tests/fuzz_cases.py (192 traces each).  LDSP_HIP_LIB selects the library build."""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import fuzz_cases
orc.build()


def collect(acc, sc, trig, ora):
    for g in ldsp._abi.SIPM_TRIG_GROUPS:
        cg, co = trig[g]["count"].cpu().numpy(), ora[g]["count"]
        same = np.nonzero(cg == co)[0]
        for f in ("x", "x_high", "x_tot", "max"):
            for r in same:
                c = int(co[r])
                if c:
                    a = trig[g][f][r][:c].cpu().numpy().astype(np.float64); b = np.asarray(ora[g][f][r][:c], dtype=np.float64)
                    d = np.abs(a - b); acc.setdefault((g, f), []).append(d[np.isfinite(d)])
    for c in ("threshold", "threshold_DC", "threshold_trap", "threshold_DC_trap"):
        a = sc[ldsp._abi.SIPM_SCALAR_COLS.index(c)].cpu().numpy().astype(np.float64); b = np.asarray(ora[c], dtype=np.float64)
        ok = np.isfinite(a) & np.isfinite(b) & (b != 0)
        acc.setdefault((c, "relative"), []).append(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))


def report(title, acc):
    print(title)
    for (g, f), v in acc.items():
        d = np.concatenate(v) if v else np.zeros(0)
        if d.size:
            print(f"  {g:18s} {f:9s} n {d.size:6d}  median {np.median(d):.3g}  p99 {np.quantile(d, 0.99):.3g}  max {d.max():.3g}")


acc = {}
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, 16384, 0.0, 16.0)
wf = ldsp.synth.sipm_batch(256, 16384, device="cuda")
sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
collect(acc, sc, trig, orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16))
report("reference configuration, 256 x 16384 traces (positions in ns, maxima in signal units, thresholds relative)", acc)
if len(sys.argv) > 2:
    nconf, seed = int(sys.argv[1]), int(sys.argv[2])
    acc = {}
    for it in range(nconf):
        L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(seed, it)
        try:
            p = ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0)
        except Exception:
            continue
        wf = fuzz_cases.sipm_traces(192, L, it, noise, mean_pulses)
        sc, trig = ldsp.sipm_run(wf, p); torch.cuda.synchronize()
        collect(acc, sc, trig, orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16))
    report(f"{nconf} randomised configurations of seed {seed}, 192 traces each", acc)
