import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 8192
cfg = ldsp.reference_test_icpc_config()
wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(n, L, device="cuda"), 0.0, 16.0)
ctx = ldsp.default_context(); ctx.enable_timing(True)
for name, fn in (("dsp_trap_rt_optimization", lambda: ldsp.dsp_trap_rt_optimization(wvfs, cfg, 500 * ldsp.us, ctx=ctx)),
                 ("dsp_trap_ft_optimization", lambda: ldsp.dsp_trap_ft_optimization(wvfs, cfg, 500 * ldsp.us, 8 * ldsp.us, ctx=ctx)),
                 ("dsp_cusp_rt_optimization", lambda: ldsp.dsp_cusp_rt_optimization(wvfs, cfg, 500 * ldsp.us, ctx=ctx)),
                 ("dsp_zac_ft_optimization", lambda: ldsp.dsp_zac_ft_optimization(wvfs, cfg, 500 * ldsp.us, 8 * ldsp.us, ctx=ctx))):
    out = fn(); torch.cuda.synchronize()
    ms = min((fn(), ctx.last_kernel_ms())[1] for _ in range(3))
    G = out.shape[0]
    b = n * (4 * L + 4 * G)
    print(f"{name}: grid {G} x {n} traces: {ms:.3f} ms -> {n/ms*1e3/1e6:.1f} Mwf/s ({n*G/ms*1e3/1e9:.2f} G filter-evaluations/s), {b/ms*1e3/1e12:.2f} TB/s = {b/ms*1e3/8e12*100:.0f}% of 8 TB/s")
import dataclasses
cfg2 = dataclasses.replace(cfg, a_grid_wl_sg=ldsp.StepRange(80 * ldsp.ns, 32 * ldsp.ns, 350 * ldsp.ns))   # from 5 points up (a cubic needs 4)
pf = {"trap": {"rt": 8 * ldsp.us, "ft": 3 * ldsp.us}}
fn = lambda: ldsp.dsp_sg_optimization(wvfs, cfg2, 500 * ldsp.us, pf, ctx=ctx)
out = fn(); torch.cuda.synchronize()
ms = min((fn(), ctx.last_kernel_ms())[1] for _ in range(3))
W = out["aoe"].shape[0]
b = n * (4 * L + 4 * (W + 4))
print(f"dsp_sg_optimization: {W} window lengths x {n} traces: {ms:.3f} ms -> {n/ms*1e3/1e6:.1f} Mwf/s, {b/ms*1e3/1e12:.2f} TB/s = {b/ms*1e3/8e12*100:.0f}% of 8 TB/s")
