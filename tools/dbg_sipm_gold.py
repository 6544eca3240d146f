import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
s = np.load("tests/golden/sipm_oracle_vectors.npz")
wf = torch.from_numpy(s["wf"]).cuda()
p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, wf.shape[1], 0.0, 16.0)
sc, trig = ldsp.sipm_run(wf, p)
for i, c in enumerate(ldsp._abi.SIPM_SCALAR_COLS):
    print(c, sc[i].cpu().numpy(), s["col__" + c])
for g in ldsp._abi.SIPM_TRIG_GROUPS:
    print(g, "count", trig[g]["count"].cpu().numpy(), s[f"trig__{g}__count"])
    for f in ("x", "x_high", "x_tot", "max"):
        a, b = trig[g][f].cpu().numpy(), s[f"trig__{g}__{f}"]
        for t in range(a.shape[0]):
            n = int(s[f"trig__{g}__count"][t])
            d = np.abs(a[t, :n] - b[t, :n])
            if n and np.nanmax(d) > 1e-3: print("   ", g, f, "trace", t, a[t, :n], b[t, :n])
