"""A/B of the QC feature kernel (k_qc_features) between library builds: rate and a checksum of the feature matrix (bit identity between builds).
Usage (GPU box): LDSP_HIP_LIB=<lib> LDSP_ALLOW_STALE=1 python tools/qc_ab.py [n]   — lengths 8192 / 4096 / 6000 / 8190 (the last: the generic path)"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import legenddsp_jl_amd as ldsp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cfg = ldsp.reference_test_icpc_config()
for L in (8192, 4096, 6000, 8190, 32768, 20002):
    m = n if L <= 8192 else n // 8
    wf = ldsp.synth.hpge_batch(m, L, device="cuda")
    w = ldsp.ArrayOfRDWaveforms(wf, 0.0, 16.0)
    for lv in (5, 2, 1):
        out = ldsp.qc_features(w, lv, cfg); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); out = ldsp.qc_features(w, lv, cfg); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        feat = out[0] if isinstance(out, (tuple, list)) else out
        h = hashlib.sha256(feat.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"L={L} levels={lv}: {best*1e3:.3f} ms -> {m/best/1e6:.1f} M waveforms/s, {m*4*L/best/1e12:.2f} TB/s of input; features sha256 {h}")
    del wf, w
