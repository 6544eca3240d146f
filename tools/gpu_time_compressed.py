"""Timing of dsp_icpc_compressed (presummed 2048 x 64 ns + windowed 3000 x 16 ns per event) and of the QC feature kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L, rate, DT = 8192, 4, 16.0
cfg = ldsp.reference_test_icpc_config()
wf = ldsp.synth.hpge_batch(n, L, device="cuda")
pre = ldsp.ArrayOfRDWaveforms(wf.view(n, L // rate, rate).sum(dim=2).contiguous(), 0.0, DT * rate)
wdw = ldsp.ArrayOfRDWaveforms(wf[:, 2000:5000].contiguous(), 2000 * DT, DT)
z = torch.zeros(n)
data = ldsp.Table(waveform_presummed=pre, waveform_windowed=wdw, presum_rate=torch.full((n,), rate, dtype=torch.int32), baseline=z,
                  timestamp=torch.arange(n), eventnumber=torch.arange(1, n + 1), daqenergy=z, t_sat_lo=z, t_sat_hi=z, deadtime=z)
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best
t = timed(lambda: ldsp.dsp_icpc_compressed(data, cfg, 500 * ldsp.us, {}))
b = n * 4 * (pre.nsamples + wdw.nsamples)
print(f"dsp_icpc_compressed n={n}: {t*1e3:.2f} ms -> {n/t/1e6:.2f} M events/s, {b/t/1e12:.3f} TB/s of input ({b/n} B/event)")
w = ldsp.ArrayOfRDWaveforms(wf, 0.0, DT)
for lv in (5, 2):
    t = timed(lambda: ldsp.qc_features(w, lv, cfg))
    print(f"qc_features levels={lv} n={n} L={L}: {t*1e3:.3f} ms -> {n/t/1e6:.1f} M waveforms/s, {n*4*L/t/1e12:.2f} TB/s = {n*4*L/t/8e12*100:.0f}% of 8 TB/s")
