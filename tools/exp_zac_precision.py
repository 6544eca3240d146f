"""Closed-form vs direct-form CUSP/ZAC on traces with a step / a second pulse / a DC offset (which feature of a pile-up
trace costs the closed form its precision).  GPU box: python tools/exp_zac_precision.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legenddsp_jl_amd as ldsp
import fuzz_cases
case = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(1, case)
p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
n = 64
base = ldsp.synth.hpge_batch(n, L, device="cuda", seed=3, noise=0.0)
bl = base[:, :1000].mean(dim=1)
other = ldsp.synth.hpge_batch(n, L, device="cuda", seed=4, noise=0.0)
second = torch.roll(other - other[:, :1], 900, dims=1) * 0.5
second_only = second.clone(); second_only[:, :900] = 0.0
step_only = torch.zeros_like(base); step_only[:, :900] = second[:, :900]
variants = {"clean": base, "+dc 3000 (ext baseline kept)": base + 3000.0, "+step in [0,900)": base + step_only,
            "+second pulse at +900": base + second_only, "+both (the pile-up rows)": base + second}
ctx = ldsp.default_context()
for name, wf in variants.items():
    res = {}
    for direct in (0, 1):
        ctx.set_option("cusp_direct", direct)
        tab = ldsp.icpc_run(wf.contiguous(), p, ctx, ext_baseline=bl.contiguous(), ext_baseline_scale=1.0)
        torch.cuda.synchronize()
        res[direct] = {k: v.cpu().numpy().astype(np.float64) for k, v in ldsp.table_columns(tab).items()}
    ctx.set_option("cusp_direct", 0)
    out = []
    for c in ("e_cusp", "e_zac", "e_cusp_max", "e_zac_max"):
        d = np.abs(res[0][c] - res[1][c])
        out.append(f"{c} max|closed-direct| {d.max():.3g} (median {np.median(d):.3g}, scale {np.abs(res[1][c]).max():.5g})")
    print(f"{name:32s} kernel={ctx.last_kernel_name()}: " + "; ".join(out))
