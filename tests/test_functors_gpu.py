"""Filter functors and extractors through the C ABI: the reference's own known-answer
values (transcribed data, files named per test) and seeded-batch parity with the oracle."""
import math

import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp

pytestmark = pytest.mark.gpu
DT = 16.0


def wv(a, t_first=0.0, dt=DT):
    a = np.atleast_2d(np.asarray(a, dtype=np.float32))
    return ldsp.ArrayOfRDWaveforms(torch.from_numpy(a).cuda(), t_first, dt)


def host(t):
    return t.cpu().numpy()


# ---- reference known answers -------------------------------------------------------
def test_haar_reference_values():            # test/test_haar_filter.jl:6-72
    step = np.concatenate([np.ones(128), 2 * np.ones(128)])
    w = wv(step, 0.0, 32000.0)
    o = ldsp.HaarAveragingFilter(2)(w)
    assert (o.t_first, o.dt, o.nsamples) == (0.0, 64000.0, 128)
    np.testing.assert_allclose(host(o.signal)[0], np.r_[np.full(64, math.sqrt(2)), np.full(64, math.sqrt(8))], rtol=2e-7)
    o2 = ldsp.HaarAveragingFilter(2)(o)
    assert (o2.dt, o2.nsamples) == (128000.0, 64)
    np.testing.assert_allclose(host(o2.signal)[0], np.r_[np.full(32, 2.0), np.full(32, 4.0)], rtol=3e-7)
    o4 = ldsp.HaarAveragingFilter(4)(w)
    assert (o4.dt, o4.nsamples) == (128000.0, 64)
    ramp = wv(np.arange(256.0), 0.0, 32000.0)
    np.testing.assert_allclose(host(ldsp.HaarAveragingFilter(2)(ramp).signal)[0], np.arange(0.5, 255, 2) * math.sqrt(2), rtol=3e-7)
    np.testing.assert_allclose(host(ldsp.HaarAveragingFilter(4)(ramp).signal)[0], np.arange(0.5, 253, 4) * math.sqrt(2), rtol=3e-7)


def test_derivative_reference_values():      # test/test_derivative.jl:6-31
    rng = np.random.default_rng(2)
    sig = rng.random(100).astype(np.float32)
    expect = np.r_[sig[1] - sig[0], np.diff(sig)]
    np.testing.assert_allclose(host(ldsp.DerivativeFilter()(wv(sig)).signal)[0], expect, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(ldsp.DerivativeFilter(0.37)(wv(sig)).signal)[0], np.float32(0.37) * expect, rtol=1e-6, atol=1e-7)


def test_moving_window_reference_values():   # test/test_moving_window.jl:6-24
    sig = np.r_[np.zeros(5), np.ones(5)]
    w = wv(sig, 0.0, 32000.0)
    np.testing.assert_allclose(host(ldsp.MovingWindowFilter(64000.0)(w).signal)[0], [0, 0, 0, 0, 0, .5, 1, 1, 1, 1], atol=1e-6)
    np.testing.assert_allclose(host(ldsp.MovingWindowMultiFilter(64000.0)(w).signal)[0], [0, 0, 0, 0, .125, .5, .875, 1, 1, 1], atol=1e-6)


def test_extremestats_reference_values():    # test/test_stats.jl:9-54
    y = np.sin(np.deg2rad(np.arange(0, 361)))
    y[[0, 180, 360]] = 0.0; y[90] = 1.0; y[270] = -1.0
    w = wv(y, 0.0, 1.0)
    e = {k: float(v[0]) for k, v in ldsp.extremestats(w).items()}
    assert (e["min"], e["max"], e["tmin"], e["tmax"]) == (-1.0, 1.0, 270.0, 90.0)
    e = {k: float(v[0]) for k, v in ldsp.extremestats(w, 0.0, 180.0).items()}
    assert (e["min"], e["max"], e["tmin"], e["tmax"]) == (0.0, 1.0, 0.0, 90.0)
    e = {k: float(v[0]) for k, v in ldsp.extremestats(w, 135.0, 225.0).items()}
    assert e["min"] == pytest.approx(-math.sqrt(0.5), abs=1e-7) and (e["tmin"], e["tmax"]) == (225.0, 135.0)
    with pytest.raises(ldsp.WindowError):    # the reference's @assert on windows outside the trace
        ldsp.extremestats(w, 0.0, 400.0)


def test_thresholdstats_mad_reference_values():   # test/test_thresholdstats.jl:7-65
    assert float(ldsp.thresholdstats_mad(wv(np.full(100, 5.0)), -10.0, 10.0)[0]) == pytest.approx(0.0, abs=1e-7)
    sym = np.r_[np.full(50, -1.0), np.full(50, 1.0)]
    assert float(ldsp.thresholdstats_mad(wv(sym), -5.0, 5.0)[0]) == pytest.approx(1.4826, abs=1e-6)
    out = np.zeros(1000); out[499:510] = 1000.0
    assert float(ldsp.thresholdstats_mad(wv(out))[0]) < 1.0
    assert float(ldsp.thresholdstats_mad(wv(np.full(100, 5.0)), 10.0, 20.0)[0]) == 0.0


def test_get_wvf_maximum_reference_values():      # test/test_interpolation.jl:6-45
    n = 100
    s = np.zeros(n); s[:4] = [1.0, 0.8, 0.5, 0.2]
    assert float(ldsp.get_wvf_maximum(wv(s), 0.0, 64.0)[0]) == 1.0
    s = np.zeros(n); s[-4:] = [0.2, 0.5, 0.8, 1.0]
    assert float(ldsp.get_wvf_maximum(wv(s), (n - 5) * DT, (n - 1) * DT)[0]) == 1.0
    s = np.zeros(n); s[49:52] = [0.5, 1.0, 0.5]
    assert float(ldsp.get_wvf_maximum(wv(s), 47 * DT, 53 * DT)[0]) == 1.0


def test_intersect_maximum_reference_values():    # test/test_intersect_maximum.jl:6-107
    n = 6200
    tend = (n - 1) * DT
    f = ldsp.IntersectMaximum(mintot=2 * DT, maxtot=100 * DT)
    s = np.zeros(n); s[1:4] = [0.5, 0.6, 0.2]
    r = f(wv(s), 0.4)
    assert int(r["multiplicity"][0]) == 1 and len(r["x"][0]) == 1
    assert float(r["x"][0][0]) == pytest.approx(12.8, abs=1e-4)
    assert float(r["x_high"][0][0]) == pytest.approx(40.0, abs=1e-4)
    assert float(r["max"][0][0]) == pytest.approx(0.6225, abs=1e-6)
    assert float(r["x_tot"][0][0]) == pytest.approx(27.2, abs=1e-4)
    s = np.zeros(n); s[-5:] = [0.3, 0.5, 0.6, 0.8, 1.0]
    r = ldsp.IntersectMaximum(mintot=2 * DT, maxtot=5 * DT)(wv(s), 0.4)
    assert int(r["multiplicity"][0]) == 1 and float(r["max"][0][0]) == 1.0
    s = np.zeros(n); s[-3:] = [0.3, 0.5, 0.6]
    r = f(wv(s), 0.4)
    assert int(r["multiplicity"][0]) == 1 and float(r["x_high"][0][0]) == pytest.approx(tend, rel=1e-7)
    s = np.zeros(n); s[99:105] = 0.8; s[199:215] = 0.9
    r = f(wv(s), 0.4)
    assert int(r["multiplicity"][0]) == 2 and float(r["x_tot"][0][1]) > float(r["x_tot"][0][0]) > 0
    r = f(wv(np.zeros(n)), 0.4)                  # no crossing: empty vectors
    assert int(r["multiplicity"][0]) == 0 and len(r["x"][0]) == 0


def test_intersect_maximum_positions_are_float64():
    """The reference's trigger positions are Float64 (src/dsp_sipm.jl:87-88 converts the time axis; the ragged columns of
    :149-156 are Vector{Float64}).  On a time axis that starts at 1e9 ns a float32 position has an ulp of 64 ns; the slabs
    are double and composed in double, so the crossing comes out to the rounding of the float32 interpolation fraction."""
    n, t0 = 4096, 1.0e9
    f = ldsp.IntersectMaximum(mintot=2 * DT, maxtot=100 * DT)
    s = np.zeros(n); s[1000:1003] = [0.5, 0.6, 0.2]; s[3000:3006] = [0.25, 0.5, 0.75, 1.0, 0.5, 0.1]
    r = f(wv(s, t0, DT), 0.4)
    assert r["x"].values.dtype == r["x_high"].values.dtype == r["x_tot"].values.dtype == torch.float64
    assert r["max"].values.dtype == torch.float32
    x, xh, xt = (host(r[k][0]) for k in ("x", "x_high", "x_tot"))
    exp_x = np.array([t0 + DT * (999 + 0.4 / 0.5), t0 + DT * (3000 + (0.4 - 0.25) / 0.25)])
    exp_h = np.array([t0 + DT * (1001 + (0.4 - 0.6) / (0.2 - 0.6)), t0 + DT * (3004 + (0.4 - 0.5) / (0.1 - 0.5))])
    np.testing.assert_allclose(x, exp_x, rtol=0, atol=2e-5)      # 16 ns x float32 epsilon of the fraction
    np.testing.assert_allclose(xh, exp_h, rtol=0, atol=2e-5)
    np.testing.assert_allclose(xt, exp_h - exp_x, rtol=0, atol=4e-5)


def test_multi_intersect_reference_values():      # test/test_multiintersect.jl:7-27
    y = np.arange(1.0, 101.0)
    w = wv(y, 1.0, 1.0)
    one = ldsp.MultiIntersect(threshold_ratios=(0.5,), mintot=1.0)(w)
    ref = ldsp.Intersect(mintot=1.0)(w, 50.0)
    assert float(one[0, 0]) == pytest.approx(float(ref["x"][0]), rel=1e-6) == pytest.approx(50.0, rel=1e-6)
    res = ldsp.MultiIntersect(threshold_ratios=tuple(np.arange(0.1, 0.95, 0.1)), mintot=1.0)(w)
    np.testing.assert_allclose(host(res)[0], np.arange(10.0, 91.0, 10.0), rtol=1e-5)


# ---- seeded-batch parity with the oracle -----------------------------------------------
@pytest.fixture(scope="module")
def batch():
    wf = ldsp.synth.hpge_batch(96, 8192, device="cuda", seed=5)
    blm = wf[:, :2000].mean(dim=1, keepdim=True)
    return ldsp.ArrayOfRDWaveforms((wf - blm).contiguous(), 0.0, DT)


def _multi_oracle(orc, x, ratios, min_n, half_n, degree, rate, dt=DT):
    """Per trace: the oracle's K crossing times, or None where the reference's boundary @assert fires."""
    res = []
    for row in x:
        try:
            res.append(orc.multi_intersect(row, ratios, min_n, half_n, degree, rate, t_first=0.0, dt=dt))
        except orc.OracleError:
            res.append(None)
    return res


@pytest.mark.parametrize("mintot_samples,half_n,degree,rate", [(2, 1, 1, 1), (3, 1, 1, 1), (4, 2, 1, 4), (5, 2, 2, 4), (6, 3, 2, 2), (40, 1, 1, 1)])
def test_multi_intersect_matches_oracle_on_hpge_traces(orc, batch, mintot_samples, half_n, degree, rate):
    """MultiIntersect with the reference's default 90 thresholds (src/multi_intersect.jl:12): the wave-parallel search
    (one wave per threshold, closed form of the reference's advance-and-rewind walk) against the oracle's serial walk, and
    against the kernel's own one-lane walk (option multi_serial) bit for bit.  mintot 40 samples takes the serial path."""
    ratios = tuple(np.arange(0.01, 0.905, 0.01))
    f = ldsp.MultiIntersect(threshold_ratios=ratios, mintot=mintot_samples * DT, n=half_n, d=degree, sampling_rate=rate)
    got = host(f(batch)).astype(np.float64)
    x = host(batch.signal).astype(np.float64)
    ora = _multi_oracle(orc, x, ratios, mintot_samples, half_n, degree, rate)
    assert all(o is not None for o in ora)
    ora = np.stack(ora)
    bad = ~((np.abs(got - ora) <= 0.02 * DT) | (np.isnan(got) & np.isnan(ora)))   # 0.02 sample: float32 interpolation of a noisy edge; NaN on both sides = a threshold that is never confirmed
    assert bad.sum() == 0, (np.argwhere(bad)[:10], got[bad][:10], ora[bad][:10])
    ctx = ldsp.default_context(batch.signal.device.index)
    ctx.set_option("multi_serial", 1)
    try:
        serial = host(f(batch))
    finally:
        ctx.set_option("multi_serial", 0)
    assert np.array_equal(serial, host(f(batch)), equal_nan=True)


def test_multi_intersect_hard_traces_and_window_status(orc):
    """Traces on which the walk does not simply climb the edge: spikes before the pulse (a confirmed early crossing of
    the low thresholds only, later thresholds cross at the edge), an initial run above the first thresholds, thresholds that
    are never confirmed (positions stay at sample 1), a negative trace (descending thresholds: serial path), a pulse
    at the very end / start (the reference's boundary @assert, src/multi_intersect.jl:75-78 -> status and WindowError)."""
    L, K = 4096, 90
    ratios = tuple(np.arange(0.01, 0.905, 0.01))
    rng = np.random.default_rng(12)
    t = np.arange(L)
    edge = 1000.0 / (1.0 + np.exp(-(t - 2000) / 40.0))
    rows = []
    for i in range(32):
        y = edge + rng.normal(0, 2.0, L)
        if i % 8 == 1:
            y[300:300 + 2 + i % 5] += 250.0           # short plateau before the pulse: low thresholds confirm there
        if i % 8 == 2:
            y[:50] += 400.0                           # the trace starts above the first thresholds
        if i % 8 == 3:
            y[2500:] = 0.0; y[2100] += 5000.0         # maximum is one spike: high thresholds never confirmed for mintot >= 2
        if i % 8 == 4:
            y = -y - 50.0                             # negative maximum: thresholds descend
        if i % 8 == 5:
            y = np.roll(y, -1990 + 3 * (i // 8))      # the edge sits at the start of the trace
        if i % 8 == 6:
            y = 1000.0 * np.exp(-t / 500.0) + rng.normal(0, 0.5, L)   # starts at its maximum: nothing is ever confirmed,
                                                                      # positions stay at sample 1 -> left boundary for n >= 2
        if i % 8 == 7:
            y = rng.normal(0, 2.0, L); y[L - 1 - i // 8:] += 1000.0   # the step arrives in the last samples: right boundary
        rows.append(y)
    x = np.asarray(rows, dtype=np.float32)
    w = wv(x)
    ctx = ldsp.default_context(w.signal.device.index)
    ctx.bind_stream()
    import ctypes as C
    from legenddsp_jl_amd import _lib
    for min_n, half_n, degree, rate in [(1, 1, 1, 1), (2, 1, 1, 1), (4, 2, 1, 2), (6, 2, 2, 4)]:
        ora = _multi_oracle(orc, x.astype(np.float64), ratios, min_n, half_n, degree, rate)
        out = torch.empty((len(rows), K), dtype=torch.float32, device="cuda")
        status = torch.empty(len(rows), dtype=torch.int32, device="cuda")
        r = np.ascontiguousarray(ratios, dtype=np.float64)
        for serial in (0, 1):
            ctx.set_option("multi_serial", serial)
            try:
                _lib.check(_lib.lib().ldsp_multi_intersect(ctx.handle, C.c_void_p(w.signal.data_ptr()), len(rows), L, 0.0, DT,
                                                           r.ctypes.data_as(C.c_void_p), K, min_n, half_n, degree, rate,
                                                           C.c_void_p(out.data_ptr()), C.c_void_p(status.data_ptr())))
            finally:
                ctx.set_option("multi_serial", 0)
            st, got = host(status), host(out).astype(np.float64)
            nerr = 0
            for i, o in enumerate(ora):
                assert (st[i] != 0) == (o is None), (min_n, half_n, serial, i, st[i])
                if o is None:
                    nerr += 1
                    continue
                ok = (np.abs(got[i] - o) <= 0.02 * DT) | (np.isnan(got[i]) & np.isnan(o))
                assert ok.all(), (min_n, half_n, serial, i, np.nonzero(~ok)[0][:8], got[i][~ok][:8], o[~ok][:8])
        assert nerr > 0 or half_n == 1                 # for n >= 2 the batch does hold window-error traces
        f = ldsp.MultiIntersect(threshold_ratios=ratios, mintot=min_n * DT, n=half_n, d=degree, sampling_rate=rate)
        if nerr > 0:
            with pytest.raises(ldsp.WindowError):
                f(w)
        else:
            f(w)


def _each(batch, fn):
    x = host(batch.signal).astype(np.float64)
    return [fn(x[i]) for i in range(x.shape[0])]


def test_filters_match_oracle(orc, batch):
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, DT)
    cases = [
        (ldsp.InvCRFilter(500 * ldsp.us), lambda x: orc.invcr(x, DT / 500e3), 2e-6),
        (ldsp.IntegratorFilter(1.0), lambda x: orc.integrator(x), 2e-6),
        (ldsp.TrapezoidalChargeFilter(10 * ldsp.us, 4 * ldsp.us), lambda x: orc.trap(x, 625, 250), 2e-5),
        (ldsp.TrapezoidalChargeFilter(40.0, 100.0, 2000.0), lambda x: orc.trap(x, 2, 6, 125), 5e-5),
        (ldsp.SavitzkyGolayFilter(100.0, 3, 1), lambda x: orc.fir(x, orc.sg_coeffs(7, 3, 1)), 2e-5),
        (ldsp.CUSPChargeFilter(5 * ldsp.us, 2.5 * ldsp.us, 1e7 * ldsp.us, 38 * ldsp.us, 2375.0), lambda x: orc.fir(x, orc.cusp_coeffs(p.cusp)), 5e-5),
        (ldsp.ZACChargeFilter(5 * ldsp.us, 2.5 * ldsp.us, 1e7 * ldsp.us, 38 * ldsp.us, 2375.0), lambda x: orc.fir(x, orc.zac_coeffs(p.zac)), 5e-5),
        (ldsp.DerivativeFilter(2.0), lambda x: orc.derivative(x, 2.0), 1e-6),
        (ldsp.HaarAveragingFilter(3), lambda x: orc.haar(x, 3), 1e-6),
        (ldsp.MovingWindowFilter(800.0), lambda x: orc.moving_window(x, 50), 2e-5),
        (ldsp.MovingWindowMultiFilter(800.0), lambda x: orc.moving_window_multi(x, 50), 5e-5),
    ]
    for flt, ofn, tol in cases:
        out = flt(batch)
        exp = np.stack(_each(batch, ofn))
        assert out.nsamples == exp.shape[1], type(flt).__name__
        scale = np.abs(exp).max()
        np.testing.assert_allclose(host(out.signal), exp, rtol=0, atol=tol * scale, err_msg=type(flt).__name__)
    tr = ldsp.TrapezoidalChargeFilter(10 * ldsp.us, 4 * ldsp.us)(batch)
    assert tr.t_first == (1500 - 1) * DT       # trailing time axis (A1)
    tt = ldsp.TruncateFilter(47 * ldsp.us, 53 * ldsp.us)(batch)
    assert (tt.nsamples, tt.t_first) == (3312 - 2938 + 1, 2938 * DT)
    np.testing.assert_array_equal(host(tt.signal), host(batch.signal)[:, 2938:3313])
    np.testing.assert_array_equal(host(ldsp.reverse_waveform(batch).signal), host(batch.signal)[:, ::-1])
    np.testing.assert_allclose(host(ldsp.multiply_waveform(batch, -1.0).signal), -host(batch.signal))


def test_extractors_match_oracle(orc, batch):
    x = host(batch.signal).astype(np.float64)
    n = x.shape[0]
    ss = ldsp.signalstats(batch, 0.0, 39 * ldsp.us)
    ts = ldsp.tailstats(batch, 70 * ldsp.us, 110 * ldsp.us)
    es = ldsp.extremestats(batch)
    sat = ldsp.saturation(batch, 0.0, 65520.0)
    for i in range(0, n, 7):
        o = orc.signalstats(x[i], 0, 2438, 0.0, DT)
        assert float(ss["mean"][i]) == pytest.approx(o["mean"], abs=2e-4)
        assert float(ss["sigma"][i]) == pytest.approx(o["sigma"], rel=2e-5)
        assert float(ss["slope"][i]) == pytest.approx(o["slope"], abs=1e-8)
        o = orc.tailstats(x[i], 4375, 6875, 0.0, DT)
        assert float(ts["τ"][i]) == pytest.approx(o["tau"], rel=1e-4)
        o = orc.extremestats(x[i], t_first=0.0, dt=DT)
        assert (float(es["max"][i]), float(es["tmax"][i])) == (pytest.approx(o["max"]), o["tmax"])
        assert float(ldsp.thresholdstats(batch, -5.0, 5.0)[i]) == pytest.approx(orc.thresholdstats(x[i], -5.0, 5.0), rel=1e-5)
    mad = host(ldsp.thresholdstats_mad(batch, -8.0, 8.0))
    for i in range(n):
        assert mad[i] == pytest.approx(orc.thresholdstats_mad(x[i], -8.0, 8.0), rel=2e-6), i
    for i in range(n):
        o = orc.saturation(x[i], 0.0, 65520.0)
        assert (int(sat["low"][i]), int(sat["high"][i]), int(sat["max_cons_low"][i]), int(sat["max_cons_high"][i])) == \
               (o["low"], o["high"], o["max_cons_low"], o["max_cons_high"])
    thr = batch.signal.amax(dim=1) * 0.5
    r = ldsp.Intersect(mintot=32.0)(batch, thr)
    est = ldsp.SignalEstimator(ldsp.PolynomialDNI(3, 700.0))(batch, r["x"] + 3000.0)
    for i in range(n):
        o = orc.intersect(x[i], float(thr[i]), 2, 0.0, DT)
        assert float(r["x"][i]) == pytest.approx(o["x"], abs=0.01) and int(r["multiplicity"][i]) == o["multiplicity"]
        assert float(est[i]) == pytest.approx(orc.signal_estimator(x[i], o["x"] + 3000.0, 44, 3, 0.0, DT), rel=2e-5)
    # unfused L3 helpers agree with the fused table
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, DT)
    raw = ldsp.synth.hpge_batch(96, 8192, device="cuda", seed=5)
    fused = ldsp.table_columns(ldsp.icpc_run(raw, p))
    pzd = ldsp.InvCRFilter(500 * ldsp.us)(ldsp.shift_waveform(ldsp.ArrayOfRDWaveforms(raw, 0.0, DT), -fused["blmean"]))
    t0 = ldsp.get_t0(pzd, 4.0)
    np.testing.assert_allclose(host(t0), host(fused["t0"]), atol=5e-4)
    t50 = ldsp.get_threshold(pzd, fused["e_max"] * 0.5, mintot=32.0)
    np.testing.assert_allclose(host(t50), host(fused["t50"]), atol=5e-4)
    qd = ldsp.get_qdrift(pzd, fused["t0"], (2500.0, 5000.0))
    np.testing.assert_allclose(host(qd), host(fused["qdrift"]), rtol=2e-4, atol=40)
    # get_intracePileUp (src/dsp_routines.jl:72-82) on the Savitzky-Golay derivative, as dsp_icpc.jl:181,189 calls it: the
    # statement-by-statement spelling through the functor entry points gives the fused table's columns
    cfg = ldsp.reference_test_icpc_config()
    sg = ldsp.SavitzkyGolayFilter(ldsp.get_fltpars({}, "sg", cfg), cfg.sg_flt_degree, 1)(pzd)
    pu = ldsp.get_intracePileUp(sg, float(cfg.inTraceCut_std_threshold), (cfg.bl_window.left, cfg.bl_window.right),
                                mintot=p.intrace_mintot * DT)
    n_f, x_f = host(fused["inTrace_n"]), host(fused["inTrace_intersect"])
    n_u, x_u = host(pu["n"]), host(pu["intersect"])
    same = n_u == n_f
    assert same.mean() >= 0.97, (n_u[~same], n_f[~same])          # a count may differ where a run barely holds (float32 sigma)
    both = same & np.isfinite(x_f)
    np.testing.assert_allclose(x_u[both], x_f[both], atol=0.6)
    assert np.array_equal(np.isnan(x_u[same]), np.isnan(x_f[same]))


def test_intersect_maximum_returns_every_crossing_beyond_the_slab(orc):
    """reference src/intersect_maximum.jl:49-56 pushes EVERY up-crossing: a trace with more crossings than the default slab
    (LDSP_MAX_TRIG = 64) comes back complete (second pass sized from the counts), multiplicity == length(x)."""
    n, L = 6, 8192
    k = np.arange(L)
    sig = np.zeros((n, L))
    sig[0] = np.sin(2 * np.pi * k / 40.0)            # ~204 crossings
    sig[1] = np.sin(2 * np.pi * k / 400.0)           # ~20: stays in the first pass
    sig[2] = np.sin(2 * np.pi * k / 25.0) * 2        # ~327
    sig[3] = 0.0                                     # none
    sig[4] = np.sin(2 * np.pi * k / 128.0)           # exactly 64 periods: 63 or 64 crossings, the boundary
    sig[5] = np.sin(2 * np.pi * k / 126.0)           # 65: one beyond
    f = ldsp.IntersectMaximum(mintot=3 * DT, maxtot=30 * DT)
    r = f(wv(sig), 0.3)
    mult = host(r["multiplicity"])
    assert mult[0] > 190 and mult[2] > 300 and mult[3] == 0 and mult[5] > 64
    for i in range(n):
        o = orc.intersect_maximum(sig[i].astype(np.float32), 0.3, 3, 30, 0.0, DT)
        assert mult[i] == o["multiplicity"]
        for fld in ("x", "x_high", "x_tot", "max"):
            got = host(r[fld][i])
            assert len(got) == mult[i]
            np.testing.assert_allclose(got, o[fld], atol=(2e-2 if fld != "max" else 1e-4), rtol=1e-5)
