"""Analytic properties of the oracle's restatement of the RadiationDetectorDSP primitives (SURVEY rows a18-a28,
"parity unpinned"): each assumption A1-A6 of DESIGN.md section 2 has consequences that do not depend on the upstream
source — unit gain on a step, exact reproduction of polynomials, cancellation of the detector's pole, zero area.
They do not replace the upstream code; they catch a restatement that contradicts its own stated model."""
import math

import numpy as np
import pytest

import legenddsp_jl_amd as ldsp


def test_invcr_cancels_the_exponential_tail(orc):
    # A5 + SURVEY a20: y = x + (dt/tau) cumsum(x).  The reference fixture's tail exp(-i/RC) (RC = tau/dt) comes out flat.
    rc = 31250.0
    i = np.arange(8192)
    x = np.where(i < 1000, 0.0, 10000.0 * np.exp(-(i - 1000) / rc))
    y = orc.invcr(x, 1.0 / rc)
    flat = y[1200:8000]
    assert np.ptp(flat) / flat.mean() < 2e-4            # first-order pole-zero cancellation: residual slope O(1/RC)
    assert flat.mean() == pytest.approx(10000.0, rel=1e-3)


def test_trapezoid_unit_gain_and_alignment(orc):
    # a21 / A1: valid mode, mean(second window) - mean(first window); a unit step comes out as 1 on the flat top,
    # output sample k covers input [k, k+flen) (trailing alignment stamps it with the LAST of those samples)
    x = np.zeros(4096); x[2000:] = 1.0
    navg, ngap = 312, 156
    o = orc.trap(x, navg, ngap)
    assert len(o) == 4096 - (2 * navg + ngap) + 1
    k0 = 2000 - (navg + ngap)            # first k whose second window lies entirely on the step ...
    assert o[k0] == pytest.approx(1.0, abs=1e-12) and o[k0 - 1] < 1.0
    assert o[2000 - navg] == pytest.approx(1.0, abs=1e-12) and o[2000 - navg + 1] < 1.0   # ... last k with window 1 entirely before it
    assert o[:2000 - (2 * navg + ngap) + 1].max() == 0.0


def test_cusp_zac_shapes(orc):
    # A4: with beta = length a unit step comes out with unit amplitude (tau practically infinite: the deconvolution
    # kernel [1, -exp(-dt/tau)] differentiates, the shape integrates back); ZAC's area-matched parabolas cost < 1 %
    p = ldsp._abi.CuspZac(312.5, 156, 2375, 1e10, 2375.0)
    hc, hz = orc.cusp_coeffs(p), orc.zac_coeffs(p)
    assert len(hc) == len(hz) == 2375
    x = np.zeros(8192); x[3000:] = 1.0
    oc, oz = orc.fir(x, hc), orc.fir(x, hz)
    assert len(oc) == len(oz) == 8192 - 2375 + 1
    assert oc.max() == pytest.approx(1.0, abs=2e-3)
    assert oz.max() == pytest.approx(1.0, abs=1e-2)
    # the maximum sits where the step is centred under the kernel's flat top (trailing alignment: output k covers [k, k+L))
    assert abs(int(np.argmax(oc)) + 2375 // 2 - 3000) <= 156 // 2 + 2
    # (a constant is NOT tested: with the deconvolution kernel applied in "same" mode — A4 — the summed taps equal the
    #  shape's edge value, -1.6e-3 for ZAC; dsp_icpc subtracts the baseline before these filters)


def test_savitzky_golay_derivative_is_exact_on_polynomials(orc):
    # A2: LSQ polynomial of degree d over an odd window, first derivative per sample at the window centre
    for npts, deg in ((7, 2), (13, 3), (5, 3)):
        c = orc.sg_coeffs(npts, deg, 1)
        i = np.arange(200.0)
        y = 0.5 + 0.25 * i - 0.01 * i ** 2 + (1e-4 * i ** 3 if deg >= 3 else 0.0)
        dy = 0.25 - 0.02 * i + (3e-4 * i ** 2 if deg >= 3 else 0.0)
        g = np.convolve(y, c, mode="valid")            # sg_coeffs are convolution (FIR) taps; the kernels hold the reversed, correlation form
        centre = (npts - 1) // 2
        np.testing.assert_allclose(g, dy[centre:centre + len(g)], rtol=1e-9, atol=1e-9)


def test_signal_estimator_reproduces_polynomials(orc):
    # A3: PolynomialDNI(d, n points): exact for polynomials of degree <= d at fractional positions, also near the edges
    i = np.arange(300.0)
    y = 3.0 - 0.2 * i + 0.003 * i ** 2 - 1e-5 * i ** 3
    f = lambda t: 3.0 - 0.2 * t + 0.003 * t ** 2 - 1e-5 * t ** 3
    for t in (0.0, 1.3, 57.25, 150.5, 298.9, 299.0):
        assert orc.signal_estimator(y, t, 44, 3, 0.0, 1.0) == pytest.approx(f(t), rel=1e-9, abs=1e-9)
    # with a time axis: position = (t - t_first) / dt
    assert orc.signal_estimator(y, 16.0 * 57.25 + 5.0, 6, 3, 5.0, 16.0) == pytest.approx(f(57.25), rel=1e-9)


def test_signalstats_of_a_line(orc):
    # A6: mean, population sigma, slope per time unit, offset at t = 0
    t0, dt = 100.0, 16.0
    i = np.arange(2500)
    y = 7.0 + 0.003 * (t0 + dt * i)
    s = orc.signalstats(y, 100, 2099, t0, dt)
    n = 2000
    assert s["slope"] == pytest.approx(0.003, rel=1e-10)
    assert s["offset"] == pytest.approx(7.0, rel=1e-9)
    assert s["mean"] == pytest.approx(y[100:2100].mean(), rel=1e-12)
    assert s["sigma"] == pytest.approx(0.003 * dt * math.sqrt((n * n - 1) / 12.0), rel=1e-9)


def test_intersect_interpolates_linearly(orc):
    # a26: first confirmed up-crossing, linear interpolation between the neighbouring samples; none -> NaN
    y = np.array([0, 0, 1, 3, 3, 3, 0, 0, 3, 3, 3, 3], dtype=float)
    r = orc.intersect(y, 2.0, 2, 10.0, 2.0)
    assert r["x"] == pytest.approx(10.0 + 2.0 * 2.5) and r["multiplicity"] == 2
    assert math.isnan(orc.intersect(y, 5.0, 2, 10.0, 2.0)["x"])


def test_sg_coefficients_equal_scipy(orc):
    """Independent implementation of assumption A2's coefficients: scipy.signal.savgol_coeffs in its convolution
    orientation (the orientation the oracle's `fir` and the HIP `ldsp_rdfilt_fir` take)."""
    from scipy.signal import savgol_coeffs
    for n, d, der in ((5, 2, 1), (7, 3, 1), (13, 3, 1), (7, 2, 0), (9, 4, 2), (3, 2, 1)):
        np.testing.assert_allclose(orc.sg_coeffs(n, d, der), savgol_coeffs(n, d, deriv=der, use="conv"), atol=1e-13)


def test_fft_form_of_the_fir_filters_equals_the_direct_form(orc):
    # The CPU baseline is timed in both forms (bench.py `cpu_baseline.forms`): the direct valid-mode convolution the
    # parity tests use, and the overlap-free FFT form a tuned CPU library would take for the 2375-tap CUSP/ZAC filters.
    # They are the same linear map: outputs agree to the rounding of a length-16384 double transform.
    rng = np.random.default_rng(11)
    x = 1000.0 + 50.0 * rng.standard_normal(8192) + np.where(np.arange(8192) > 3000, 9000.0, 0.0)
    h = orc.cusp_coeffs(ldsp._abi.CuspZac(312.5, 156, 2375, 1e10, 2375.0))
    try:
        orc.set_fir_mode(False)
        direct = orc.fir(x, h)
        orc.set_fir_mode(True)
        viafft = orc.fir(x, h)
    finally:
        orc.set_fir_mode(False)
    assert direct.shape == viafft.shape == (8192 - len(h) + 1,)
    assert np.abs(direct - viafft).max() <= 1e-9 * np.abs(direct).max()
    # and through the whole chain: same table
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, 16.0)
    wf = ldsp.synth.hpge_batch(8, 8192, seed=3).numpy()
    try:
        a = orc.dsp_icpc(wf, p)
        orc.set_fir_mode(True)
        b = orc.dsp_icpc(wf, p)
    finally:
        orc.set_fir_mode(False)
    for k in a:
        np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-9, equal_nan=True, err_msg=k)
