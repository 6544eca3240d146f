import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
from legenddsp_jl_amd import compressed as C
n, L, rate, DT = 65536, 8192, 4, 16.0
cfg = ldsp.reference_test_icpc_config()
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
pre = ldsp.ArrayOfRDWaveforms(wf.view(n, L // rate, rate).sum(dim=2).contiguous(), 0.0, DT * rate)
wdw = ldsp.ArrayOfRDWaveforms(wf[:, 2000:5000].contiguous(), 2000 * DT, DT)
tau = 500 * ldsp.us
def T(f, reps=5):
    f(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b * 1e3
pa = ldsp.lower_icpc(cfg, tau, {}, pre.nsamples, pre.t_first, pre.dt, presum_rate=rate)
print("lower_icpc pre      %.3f ms" % T(lambda: ldsp.lower_icpc(cfg, tau, {}, pre.nsamples, pre.t_first, pre.dt, presum_rate=rate)))
print("fused pre           %.3f ms" % T(lambda: ldsp.icpc_run(pre.signal, pa)))
A = ldsp.table_columns(ldsp.icpc_run(pre.signal, pa))
print("4 x signalstats     %.3f ms" % T(lambda: [ldsp.signalstats(pre, w.left, w.right) for w in (cfg.auxbl1_window, cfg.auxbl2_window, cfg.auxpz1_window, cfg.auxpz2_window)]))
print("windowed (fused)    %.3f ms" % T(lambda: C.windowed_columns(wdw, A["blmean"], rate, cfg, tau, {})))
print("lower windowed      %.3f ms" % T(lambda: ldsp.lower_icpc(cfg, tau, {}, wdw.nsamples, wdw.t_first, wdw.dt, windowed=True)))
z = torch.zeros(n)
data = ldsp.Table(waveform_presummed=pre, waveform_windowed=wdw, presum_rate=torch.full((n,), rate, dtype=torch.int32), baseline=z,
                  timestamp=torch.arange(n), eventnumber=torch.arange(1, n + 1), daqenergy=z, t_sat_lo=z, t_sat_hi=z, deadtime=z)
print("whole routine       %.3f ms" % T(lambda: ldsp.dsp_icpc_compressed(data, cfg, tau, {})))
