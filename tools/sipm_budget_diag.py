"""tools/sipm_budget_diag.py [it ...] — for the randomised dsp_sipm configurations of tests/fuzz_cases.py: every trigger whose position differs from
the oracle's by more than 0.01 ns, with what tests/sipm_budget.py allows it, the slope and level at the crossing and the difference of the
two thresholds (n_sigma x MAD) — which of them explains the difference."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
from oracle import oracle as orc
import fuzz_cases, sipm_budget
seed = int(os.environ.get("FUZZ_SEED", "1"))
its = [int(a) for a in sys.argv[1:]] or list(range(6))
n = 192
for it in its:
    L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(seed, it)
    p = ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0)
    wf = fuzz_cases.sipm_traces(n, L, it, noise, mean_pulses)
    sc, trig = ldsp.sipm_run(wf, p)
    torch.cuda.synchronize()
    host = wf.cpu().numpy()
    ora = orc.dsp_sipm(host, p, nthreads=16)
    print(f"== case {it}: {descr}")
    for g in ldsp._abi.SIPM_TRIG_GROUPS:
        cg, co = trig[g]["count"].cpu().numpy(), ora[g]["count"]
        col, nsname = sipm_budget.THR_COL[g]
        thr_g = sc[ldsp._abi.SIPM_SCALAR_COLS.index(col)].cpu().numpy().astype(np.float64)
        ns = float(getattr(p, nsname))
        for r in np.nonzero((cg == co) & (co > 0))[0]:
            c = int(co[r])
            xa = trig[g]["x"][r][:c].cpu().numpy().astype(np.float64); xb = ora[g]["x"][r][:c]
            ha = trig[g]["x_high"][r][:c].cpu().numpy().astype(np.float64); hb = ora[g]["x_high"][r][:c]
            if max(np.abs(xa - xb).max(), np.abs(ha - hb).max()) <= 0.01:
                continue
            sig = sipm_budget.signals64(host[r], p, orc)
            s, t0 = sipm_budget.group_signal(g, sig, p)
            b = sipm_budget.trigger_budgets(g, host[r], p, orc, {"x": xb, "x_high": hb}, dth=ns * (thr_g[r] - ora[col][r]))
            for j in range(c):
                for nm, a_, b_, bud in (("x", xa[j], xb[j], b["x"][j]), ("x_high", ha[j], hb[j], b["x_high"][j])):
                    if abs(a_ - b_) > 0.01:
                        k = int(np.ceil((b_ - t0) / p.dt)); slope = s[k] - s[k - 1]
                        dth = ns * (thr_g[r] - ora[col][r])
                        print(f"  {g} row {r} trig {j} {nm}: gpu-oracle {a_ - b_:+.4f} ns  budget {bud:.4f}  slope {slope:+.3e}/sample  level {max(abs(s[k]), abs(s[k-1])):.3e}"
                              f"  n_sigma*(thr_gpu - thr_ora) {dth:+.3e} -> moves the crossing by {-dth / slope * p.dt if slope else float('nan'):+.4f} ns"
                              f"  thr {ora[col][r]:.6f}")
