#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ (run from the repo root: `python tests/golden/make_golden.py`).

The reference is Julia and cannot run in this image, and most of its arithmetic lives in an
un-vendored dependency (RadiationDetectorDSP.jl), so there are two kinds of vectors:

* `reference_known_answers.json` - DATA transcribed from the reference's own test files: the inputs
  its tests build and the outputs they assert (file:line given per case).  These pin the oracle
  (tests/test_oracle_golden.py) and are checked directly against the HIP entry points
  (tests/test_golden_gpu.py).
* `icpc_oracle_vectors.npz`, `sipm_oracle_vectors.npz` - inputs (float32 traces, the reference test
  configs of test/test_dsp_icpc.jl:50-161 and test/test_dsp_sipm.jl) and the outputs of the pinned
  CPU oracle (oracle/ldsp_oracle.c, float64).  They freeze the oracle (a CPU test fails if a change
  to oracle/ moves any value) and give the GPU tests expected vectors that do not depend on building
  oracle/ on the GPU box.  Where the oracle's restatement rests on assumptions A1-A7
  these vectors are only as good as those assumptions: "parity unpinned" for them.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import legenddsp_jl_amd as ldsp  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def known_answers():
    """Each case: `op` + arguments, the input the reference test builds (`x`, or `x_sparse` = zeros with a
    few values set; first time `t0`, step `dt`, ns), and what the test asserts: `expect[field]` is a number /
    list (equality within `atol`/`rtol`) or a dict of bounds (ge / gt / le / lt), `relations` are the
    cross-field assertions, written over the result fields."""
    s2, s8 = math.sqrt(2.0), math.sqrt(8.0)
    DT = 16.0
    C = []
    # ---- test/test_haar_filter.jl  (32 us sampling; output step = ds * 32 us, length ceil(L/ds))
    step = [1.0] * 128 + [2.0] * 128
    ramp = [float(i) for i in range(256)]
    C.append(dict(ref="test/test_haar_filter.jl:14-21", op="haar", ds=2, t0=0.0, dt=32000.0, x=step, expect=[s2] * 64 + [s8] * 64,
                  expect_dt=64000.0, rtol=3e-7))
    C.append(dict(ref="test/test_haar_filter.jl:22-26", op="haar", ds=2, t0=0.0, dt=64000.0, x=[s2] * 64 + [s8] * 64,
                  expect=[2.0] * 32 + [4.0] * 32, expect_dt=128000.0, rtol=4e-7))
    C.append(dict(ref="test/test_haar_filter.jl:29-36", op="haar", ds=4, t0=0.0, dt=32000.0, x=step, expect=[s2] * 32 + [s8] * 32,
                  expect_dt=128000.0, rtol=3e-7))
    C.append(dict(ref="test/test_haar_filter.jl:44-51", op="haar", ds=2, t0=0.0, dt=32000.0, x=ramp,
                  expect=[(0.5 + 2 * i) * s2 for i in range(128)], expect_dt=64000.0, rtol=3e-7))
    C.append(dict(ref="test/test_haar_filter.jl:52-56", op="haar", ds=2, t0=0.0, dt=64000.0, x=[(0.5 + 2 * i) * s2 for i in range(128)],
                  expect=[3.0 + 8 * i for i in range(64)], expect_dt=128000.0, rtol=1e-6))
    C.append(dict(ref="test/test_haar_filter.jl:59-66", op="haar", ds=4, t0=0.0, dt=32000.0, x=ramp,
                  expect=[(0.5 + 4 * i) * s2 for i in range(64)], expect_dt=128000.0, rtol=3e-7))
    # ---- test/test_moving_window.jl  (32 us sampling, 64 us window = 2 samples)
    sig = [0.0] * 5 + [1.0] * 5
    C.append(dict(ref="test/test_moving_window.jl:6-15", op="moving_window", length=64000.0, t0=0.0, dt=32000.0, x=sig,
                  expect=[0, 0, 0, 0, 0, 0.5, 1, 1, 1, 1], atol=1e-6))
    C.append(dict(ref="test/test_moving_window.jl:16-24", op="moving_window_multi", length=64000.0, t0=0.0, dt=32000.0, x=sig,
                  expect=[0, 0, 0, 0, 0.125, 0.5, 0.875, 1, 1, 1], atol=1e-6))
    # ---- test/test_derivative.jl: out = gain * vcat(sig[2]-sig[1], diff(sig)) for a random signal (rand(100); seeded here)
    rng = np.random.default_rng(2)
    rs = [float(np.float32(v)) for v in rng.random(100)]
    d = [rs[1] - rs[0]] + [rs[i] - rs[i - 1] for i in range(1, 100)]
    C.append(dict(ref="test/test_derivative.jl:8-13", op="derivative", gain=1.0, t0=0.0, dt=DT, x=rs, expect=d, rtol=1e-6, atol=1e-7))
    C.append(dict(ref="test/test_derivative.jl:15-18", op="derivative", gain=0.37, t0=0.0, dt=DT, x=rs, expect=[0.37 * v for v in d],
                  rtol=1e-6, atol=1e-7))
    # ---- test/test_interpolation.jl  (100 samples at 16 ns; windows in ns)
    n = 100
    C.append(dict(ref="test/test_interpolation.jl:11-21", op="get_wvf_maximum", start=0.0, stop=64.0, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=0, values=[1.0, 0.8, 0.5, 0.2]), expect=dict(ge=1.0, lt=1.1)))
    C.append(dict(ref="test/test_interpolation.jl:23-33", op="get_wvf_maximum", start=(n - 5) * DT, stop=(n - 1) * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=n - 4, values=[0.2, 0.5, 0.8, 1.0]), expect=dict(ge=1.0, lt=1.1)))
    C.append(dict(ref="test/test_interpolation.jl:35-44", op="get_wvf_maximum", start=47 * DT, stop=53 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=49, values=[0.5, 1.0, 0.5]), expect=dict(ge=1.0, lt=1.2)))
    # ---- test/test_intersect_maximum.jl  (6200 samples at 16 ns, threshold 0.4, mintot = 2 samples)
    n = 6200
    tl = (n - 1) * DT   # times[end]
    C.append(dict(ref="test/test_intersect_maximum.jl:12-31", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=100 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=1, values=[0.5, 0.6, 0.2]),
                  expect=dict(multiplicity=1, n_x=1, n_max=1, x=[dict(gt=0.0, lt=48.0)], max=[dict(ge=0.6, lt=0.7)], x_tot=[dict(gt=0.0)]),
                  relations=["x_high[0] > x[0]", "abs(x_tot[0] - (x_high[0] - x[0])) < 1e-3"]))
    C.append(dict(ref="test/test_intersect_maximum.jl:33-50", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=100 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=n - 3, values=[0.5, 0.6, 0.2]),
                  expect=dict(multiplicity=1, n_x=1, n_max=1, x=[dict(gt=tl - 3 * DT, lt=tl)], max=[dict(ge=0.6, lt=0.7)], x_tot=[dict(gt=0.0)]),
                  relations=["x_high[0] > x[0]"]))
    C.append(dict(ref="test/test_intersect_maximum.jl:52-65", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=5 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=n - 5, values=[0.3, 0.5, 0.6, 0.8, 1.0]), expect=dict(multiplicity=1, max=[1.0]), atol=0))
    C.append(dict(ref="test/test_intersect_maximum.jl:67-79", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=100 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=n - 3, values=[0.3, 0.5, 0.6]),
                  expect=dict(multiplicity=1, x=[dict(gt=tl - 3 * DT)], x_high=[tl]), atol=1e-2))
    C.append(dict(ref="test/test_intersect_maximum.jl:81-90", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=100 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=0, at=0, values=[]), expect=dict(multiplicity=0, n_x=0, n_x_high=0, n_x_tot=0, n_max=0)))
    C.append(dict(ref="test/test_intersect_maximum.jl:92-106", op="intersect_maximum", threshold=0.4, mintot=2 * DT, maxtot=100 * DT, t0=0.0, dt=DT,
                  x_sparse=dict(n=n, at=99, values=[0.8] * 6 + [0.0] * 94 + [0.9] * 16),
                  expect=dict(multiplicity=2, n_x_high=2, n_x_tot=2, x_tot=[dict(gt=0.0), dict(gt=0.0)]), relations=["x_tot[1] > x_tot[0]"]))
    # ---- test/test_multiintersect.jl  (y = 1..100 on x = 1..100; thresholds = ratio * maximum)
    y = [float(i) for i in range(1, 101)]
    C.append(dict(ref="test/test_multiintersect.jl:8-13", op="multi_intersect", ratios=[0.5], mintot=1.0, t0=1.0, dt=1.0, x=y, expect=[50.0], rtol=1e-6,
                  note="the test asserts equality with Intersect(mintot)(wvf, 0.5*100).x; on this ramp that crossing is x = 50"))
    C.append(dict(ref="test/test_multiintersect.jl:15-25", op="multi_intersect", ratios=[round(0.1 * i, 10) for i in range(1, 10)], mintot=1.0,
                  t0=1.0, dt=1.0, x=y, expect=[10.0 * i for i in range(1, 10)], rtol=1e-5))
    # ---- test/test_stats.jl  (sind per degree, exact at the quadrant points; time = index)
    sine = [math.sin(math.radians(i)) for i in range(361)]
    sine[0] = sine[180] = sine[360] = 0.0; sine[90] = 1.0; sine[270] = -1.0
    sine[135] = sine[45]; sine[225] = -sine[45]
    C.append(dict(ref="test/test_stats.jl:11-18", op="extremestats", t0=0.0, dt=1.0, x=sine,
                  expect=dict(min=-1.0, max=1.0, tmin=270.0, tmax=90.0), atol=0))
    C.append(dict(ref="test/test_stats.jl:19-24", op="extremestats", start=0.0, stop=180.0, t0=0.0, dt=1.0, x=sine,
                  expect=dict(min=0.0, max=1.0, tmin=0.0, tmax=90.0), atol=0))
    C.append(dict(ref="test/test_stats.jl:25-30", op="extremestats", start=135.0, stop=225.0, t0=0.0, dt=1.0, x=sine,
                  expect=dict(min=-math.sqrt(0.5), max=math.sqrt(0.5), tmin=225.0, tmax=135.0), atol=1e-7))
    # ---- test/test_thresholdstats.jl
    C.append(dict(ref="test/test_thresholdstats.jl:11-16", op="thresholdstats_mad", lo=-10.0, hi=10.0, t0=0.0, dt=DT, x=[5.0] * 100, expect=0.0, atol=1e-7))
    C.append(dict(ref="test/test_thresholdstats.jl:20-26", op="thresholdstats_mad", lo=-5.0, hi=5.0, t0=0.0, dt=DT, x=[-1.0] * 50 + [1.0] * 50,
                  expect=1.4826, atol=1e-6))
    C.append(dict(ref="test/test_thresholdstats.jl:30-38", op="thresholdstats_mad", lo=None, hi=None, t0=0.0, dt=DT,
                  x_sparse=dict(n=1000, at=499, values=[1000.0] * 11), expect=dict(lt=1.0)))
    C.append(dict(ref="test/test_thresholdstats.jl:42-47", op="thresholdstats_mad", lo=10.0, hi=20.0, t0=0.0, dt=DT, x=[5.0] * 100, expect=0.0, atol=1e-7))
    return C


def main():
    orc.build()
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
        json.dump(dict(note="inputs and asserted outputs transcribed from the reference's test files (data, not code)",
                       cases=known_answers()), f)

    # dsp_icpc: 6 seeded synthetic traces + the reference's noiseless fixture waveform + a saturated + a pile-up trace
    L = 8192
    wf = ldsp.synth.hpge_batch(6, L, device="cpu").numpy().astype(np.float32)
    ref_wf = ldsp.synth.reference_hpge_waveform().float().numpy()[None]
    sat = wf[:1].copy(); sat[0, 3000:3040] = 65535.0; sat[0, 100:104] = 0.0
    dbl = wf[1:2].copy(); dbl[0, 4500:] += wf[2, 2600:2600 + L - 4500] - wf[2, :2000].mean()   # in-trace pile-up
    wf = np.concatenate([wf, ref_wf, sat, dbl]).astype(np.float32)
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
    out = orc.dsp_icpc(wf, p, nthreads=8, strict=False)
    cols = list(ldsp._abi.ICPC_COLS)
    table = np.stack([np.asarray(out[c], np.float64) for c in cols], 1)
    pz = orc.icpc_pz_trap(wf, p)
    np.savez_compressed(os.path.join(HERE, "icpc_oracle_vectors.npz"), wf=wf, columns=np.array(cols), table=table,
                        pz_blmean=np.asarray(pz["blmean"], np.float64), pz_e10410=np.asarray(pz["e_10410"], np.float64),
                        config=np.array("reference_test_icpc_config, tau = 500 us, t_first = 0, dt = 16 ns"))

    # dsp_sipm: 4 seeded traces at the reference fixture's length 6250 (L % 4 != 0).  The reference's own noiseless
    # fixture pulse is left out: its MAD thresholds are exactly 0, so every trigger decision is a comparison of
    # rounding noise with 0 (its smoke properties are asserted in tests/test_sipm_gpu.py instead).
    Ls = 6250
    ws = ldsp.synth.sipm_batch(4, Ls, device="cpu", seed=9).numpy().astype(np.float32)
    ps = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, Ls, 0.0, 16.0)
    so = orc.dsp_sipm(ws, ps, nthreads=4)
    flat_out = {}
    for c in ldsp._abi.SIPM_SCALAR_COLS:
        flat_out["col__" + c] = np.asarray(so[c], np.float64)
    for g in ldsp._abi.SIPM_TRIG_GROUPS:
        for f in ("count", "x", "x_high", "x_tot", "max"):
            flat_out[f"trig__{g}__{f}"] = np.asarray(so[g][f])
    np.savez_compressed(os.path.join(HERE, "sipm_oracle_vectors.npz"), wf=ws,
                        config=np.array("reference_test_sipm_config + pars_optimization sg.wl = 200 ns, t_first = 0, dt = 16 ns"), **flat_out)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
