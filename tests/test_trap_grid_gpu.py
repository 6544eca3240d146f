"""SURVEY 8(f) row 1, trapezoid part: the filter-optimisation grid scans against the oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp

pytestmark = pytest.mark.gpu
L = 8192


def _oracle(orc, wvfs, p, traps, offsets):
    return orc.trap_grid(wvfs.signal.cpu().numpy(), p, traps, offsets)


def test_trap_rt_optimization_matches_oracle(orc):
    cfg = ldsp.reference_test_icpc_config()
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(48, L, device="cuda", seed=31), 0.0, 16.0)
    grid = list(cfg.e_grid_rt_trap)
    out = ldsp.dsp_trap_rt_optimization(wvfs, cfg, 500 * ldsp.us, ft=2 * ldsp.us).cpu().numpy()
    assert out.shape == (len(grid), 48)
    p = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 0, cfg.enc_pickoff_trap)
    traps = [ldsp.config.trap_samples(rt, 2 * ldsp.us, 16.0) for rt in grid]
    ora = _oracle(orc, wvfs, p, traps, None)
    # the pick-off sits on the baseline side of the pulse for most grid values (ENC scan): values of a few ADC
    np.testing.assert_allclose(out, ora, rtol=2e-5, atol=0.05)


def test_trap_ft_optimization_matches_oracle(orc):
    cfg = ldsp.reference_test_icpc_config()
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(48, L, device="cuda", seed=32), 0.0, 16.0)
    rt = 8 * ldsp.us
    grid = list(cfg.e_grid_ft_trap)
    out = ldsp.dsp_trap_ft_optimization(wvfs, cfg, 500 * ldsp.us, rt).cpu().numpy()
    p = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 1)
    traps = [ldsp.config.trap_samples(rt, ft, 16.0) for ft in grid]
    offsets = [rt + ft / 2 for ft in grid]
    ora = _oracle(orc, wvfs, p, traps, offsets)
    np.testing.assert_allclose(out, ora, rtol=2e-5, atol=0.05)
    # energies on the flat top: close to the full chain's e_trap scale (amplitude of the pulses)
    assert np.all(out > 100.0)


def test_grid_argument_checks():
    cfg = ldsp.reference_test_icpc_config()
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(4, L, device="cuda"), 0.0, 16.0)
    p = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 0, cfg.enc_pickoff_trap)
    with pytest.raises(ldsp.LdspError):   # trapezoid longer than the trace
        ldsp.trap_grid_run(wvfs.signal, p, [ldsp._abi.Trap(5000, 10, 5000)])
    with pytest.raises(ldsp.LdspError):   # more grid points than the kernel holds
        ldsp.trap_grid_run(wvfs.signal, p, [ldsp._abi.Trap(10, 2, 10)] * (ldsp._abi.LDSP_MAX_GRID + 1))


@pytest.mark.parametrize("kind", ["cusp", "zac"])
def test_cusp_zac_grid_scans_match_oracle(orc, kind):
    """dsp_{cusp,zac}_rt_optimization and _ft_optimization (direct-form FIR at the estimator window) vs the oracle."""
    cfg = ldsp.reference_test_icpc_config()
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(12, L, device="cuda", seed=41), 0.0, 16.0)
    length = getattr(cfg, f"flt_length_{kind}")
    # rise-time grid at a fixed pick-off (ENC scan)
    grid = list(getattr(cfg, f"e_grid_rt_{kind}"))[::6]            # a subset keeps the oracle's direct convolutions short
    p0 = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 0, getattr(cfg, f"enc_pickoff_{kind}"))
    taps = ldsp.cuspzac_grid_taps(kind, [(rt, 2 * ldsp.us) for rt in grid], length, 16.0)
    out = ldsp.fir_grid_run(wvfs.signal, p0, taps).cpu().numpy()
    ora = orc.fir_grid(wvfs.signal.cpu().numpy(), p0, taps)
    np.testing.assert_allclose(out, ora, rtol=3e-5, atol=0.1)
    # flat-top grid at t50 + length/2 (energy scan) through the routine itself
    fn = ldsp.dsp_cusp_ft_optimization if kind == "cusp" else ldsp.dsp_zac_ft_optimization
    full = fn(wvfs, cfg, 500 * ldsp.us, 8 * ldsp.us).cpu().numpy()
    gft = list(getattr(cfg, f"e_grid_ft_{kind}"))
    assert full.shape == (len(gft), 12)
    p1 = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 1)
    sel = [0, len(gft) // 2, len(gft) - 1]
    taps1 = ldsp.cuspzac_grid_taps(kind, [(8 * ldsp.us, gft[j]) for j in sel], length, 16.0)
    ora1 = orc.fir_grid(wvfs.signal.cpu().numpy(), p1, taps1, [length / 2] * len(sel))
    np.testing.assert_allclose(full[sel], ora1, rtol=3e-5, atol=0.1)
    assert np.all(full > 100.0)   # energies of the pulses


def test_sg_optimization_matches_oracle(orc):
    """dsp_sg_optimization: A/E over the Savitzky-Golay window-length grid, energy, t50, baseline statistics."""
    import dataclasses
    # (the fixture's grid starts at 30 ns = 3 points, fewer than a cubic needs: start at 80 ns = 5 points)
    cfg = dataclasses.replace(ldsp.reference_test_icpc_config(), a_grid_wl_sg=ldsp.StepRange(80 * ldsp.ns, 32 * ldsp.ns, 350 * ldsp.ns))
    n = 24
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(n, L, device="cuda", seed=51), 0.0, 16.0)
    pf = {"trap": {"rt": 8 * ldsp.us, "ft": 3 * ldsp.us}}
    res = ldsp.dsp_sg_optimization(wvfs, cfg, 500 * ldsp.us, pf)
    grid = list(cfg.a_grid_wl_sg)
    assert res["aoe"].shape == (len(grid), n) and bool((res["qc_label"] == -1).all())
    p = ldsp.lower_trap_grid(cfg, 500 * ldsp.us, L, 0.0, 16.0, 1)
    npts = [ldsp.config.sg_npoints(wl, 16.0) for wl in grid]
    frm = [ldsp.config.window_index(cfg.current_window.left, (m - 1) * 16.0, 16.0) for m in npts]
    until = [ldsp.config.window_index(cfg.current_window.right, (m - 1) * 16.0, 16.0) for m in npts]
    tr = ldsp.config.trap_samples(8 * ldsp.us, 3 * ldsp.us, 16.0)
    ora = orc.sg_optimization(wvfs.signal.cpu().numpy(), p, tr, 8 * ldsp.us + 1.5 * ldsp.us, npts, int(cfg.sg_flt_degree), frm, until)
    np.testing.assert_allclose(res["energy"].cpu().numpy(), ora["energy"], rtol=2e-5, atol=0.05)
    np.testing.assert_allclose(res["t50"].cpu().numpy(), ora["t50_us"], atol=5e-4)
    np.testing.assert_allclose(res["blmean"].cpu().numpy(), ora["blmean"], atol=2e-3)
    np.testing.assert_allclose(res["blslope"].cpu().numpy(), ora["blslope"], atol=1e-8, rtol=1e-3)
    aoe = ora["amax"] / ora["energy"][None, :]
    np.testing.assert_allclose(res["aoe"].cpu().numpy(), aoe, rtol=3e-4, atol=1e-6)


def test_qc_and_qdrift_flt_optimization_agree_with_the_fused_chain():
    """dsp_qc_flt_optimization (no classifier) and dsp_qdrift_flt_optimization reproduce what the fused dsp_icpc kernel
    computes for the same definitions: blmean / blslope / t50 and the qdrift column."""
    cfg = ldsp.reference_test_icpc_config()
    n = 64
    sig = ldsp.synth.hpge_batch(n, L, device="cuda", seed=61)
    wvfs = ldsp.ArrayOfRDWaveforms(sig, 0.0, 16.0)
    p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, 16.0)
    full = {k: v for k, v in ldsp.table_columns(ldsp.icpc_run(sig, p)).items()}
    qc = ldsp.dsp_qc_flt_optimization(wvfs, cfg, 500 * ldsp.us)
    torch.testing.assert_close(qc["blmean"], full["blmean"], rtol=0, atol=0)          # same summation, same pivot
    torch.testing.assert_close(qc["blslope"], full["blslope"], rtol=1e-4, atol=2e-11)   # the lean kernel finishes the window in float32 (rcp / sqrt)
    # t50 here crosses half the maximum of the pole-zero corrected trace (:46), dsp_icpc half the raw maximum minus
    # baseline (dsp_icpc.jl:111,133): a few ns apart, and the flat-top energy follows within 1e-3
    torch.testing.assert_close(qc["t50"], full["t50"], rtol=0, atol=0.05)
    torch.testing.assert_close(qc["energy"], full["e_trap"], rtol=1e-3, atol=0.5)      # default trapezoid = trap_opt without pars_filter
    assert bool((qc["qc_label"] == -1).all())
    qd = ldsp.dsp_qdrift_flt_optimization(wvfs, full["blmean"], cfg, 500 * ldsp.us)
    torch.testing.assert_close(qd, full["qdrift"], rtol=2e-4, atol=30.0)               # unfused spelling: float32 cumsums over 8192 samples


def test_thin_routines_dsp_decay_times_and_dsp_puls(orc):
    """recombinations of the path's stages through the functor entry points."""
    cfg = ldsp.reference_test_icpc_config()
    n = 32
    sig = ldsp.synth.hpge_batch(n, L, device="cuda", seed=71)
    wvfs = ldsp.ArrayOfRDWaveforms(sig, 0.0, 16.0)
    p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, 16.0)
    full = ldsp.table_columns(ldsp.icpc_run(sig, p))
    tau = ldsp.dsp_decay_times(wvfs, cfg)
    torch.testing.assert_close(tau, full["tail_tau"] / 1000.0, rtol=2e-5, atol=1e-3)      # us; the fused kernel reports time-axis units
    data = ldsp.Table(waveform=wvfs, baseline=torch.zeros(n), timestamp=torch.arange(n), eventnumber=torch.arange(1, n + 1), daqenergy=torch.zeros(n))
    r = ldsp.dsp_puls(data, cfg)
    for c in ("blmean", "blsigma", "blslope", "bloffset"):
        torch.testing.assert_close(r[c], full[c], rtol=2e-5, atol=2e-3 if c != "blslope" else 1e-8)
    x = sig.cpu().numpy().astype(np.float64)
    xs = x - full["blmean"].cpu().numpy()[:, None].astype(np.float64)
    np.testing.assert_allclose(r["e_max"].cpu().numpy(), xs.max(axis=1), rtol=1e-6, atol=2e-3)
    e_ref = np.array([orc.trap(xs[i], 625, 250).max() for i in range(n)])
    np.testing.assert_allclose(r["e_10410"].cpu().numpy(), e_ref, rtol=2e-5, atol=0.05)
    t_ref = np.array([orc.intersect(xs[i], 0.5 * xs[i].max(), ldsp.config.nsamples(1000.0, 16.0), 0.0, 16.0)["x"] for i in range(n)]) / 1000.0
    np.testing.assert_allclose(r["t50"].cpu().numpy(), np.nan_to_num(t_ref), atol=5e-4)
    assert torch.equal(r["eventID_fadc"], torch.arange(1, n + 1))


def test_thin_routine_dsp_pmts(orc):
    """dsp_pmts (reference src/dsp_pmts.jl:3-65) through the functor entry points, against the oracle's functors."""
    n, Lp, dt = 24, 1024, 4.0
    g = torch.Generator().manual_seed(3)
    sig = 100.0 + torch.randn(n, Lp, generator=g) * 0.8
    for i in range(n):                                   # a few negative-going-then-positive pulses of 12..40 samples
        for k in range(1 + i % 3):
            s = 200 + 250 * k + 7 * i
            sig[i, s:s + 12 + 4 * k] += 30.0 + 5.0 * k
    sig[0, 900:905] = 4095.0                             # saturated samples
    wv = ldsp.ArrayOfRDWaveforms(sig.cuda(), 0.0, 16.0)   # the stored step is replaced by time_axis_step_length
    cfg = dict(time_axis_step_length=dt, baseline_window_start=0.0, baseline_window_end=600.0, min_tot_intersect=8.0,
               max_tot_intersect=400.0, intersect_threshold=10.0, wsg_window_length=28.0, wsg_flt_degree=2, wsg_weight=0,
               saturation_limit_high=4095.0, saturation_limit_low=0.0)
    data = ldsp.Table(waveform=wv, timestamp=torch.arange(n), channel=torch.full((n,), 7), eventnumber=torch.arange(1, n + 1), daqenergy=torch.zeros(n))
    r = ldsp.dsp_pmts(data, cfg)
    assert r.columnnames == ["timestamp", "eventID_fadc", "e_fc", "channel", "raw_pulse_height", "raw_pulse_low", "raw_t0_hi", "raw_t0_low",
                             "trig_max", "trig_t", "trig_mult", "sat_low", "sat_high", "pulse_height", "pulse_low", "t0_hi", "t0_low",
                             "bl_mean", "bl_sigma", "bl_slope"]
    x = sig.numpy().astype(np.float64)
    h = orc.sg_coeffs(7, 2, 0)
    for i in range(n):
        st = orc.signalstats(x[i], 0, 150, 0.0, dt)
        assert float(r["bl_mean"][i]) == pytest.approx(st["mean"], abs=2e-4) and float(r["bl_sigma"][i]) == pytest.approx(st["sigma"], rel=1e-4)
        xs = x[i] - st["mean"]
        assert float(r["raw_pulse_height"][i]) == pytest.approx(xs.max(), abs=2e-3)
        assert float(r["raw_t0_hi"][i]) == pytest.approx(dt * int(np.argmax(xs)), abs=1e-3)
        o = orc.intersect_maximum(xs, 10.0, 2, 100, 0.0, dt)
        assert int(r["trig_mult"][i]) == o["multiplicity"] == 1 + i % 3 + (i == 0)   # trace 0: + the saturated burst
        np.testing.assert_allclose(r["trig_t"][i].cpu().numpy(), o["x"], atol=0.02)
        np.testing.assert_allclose(r["trig_max"][i].cpu().numpy(), o["max"], atol=2e-3)
        sm = orc.fir(xs, h)
        assert float(r["pulse_height"][i]) == pytest.approx(sm.max(), abs=2e-3)
    assert int(r["sat_high"][0]) == 5 and int(r["sat_high"][1:].sum()) == 0
    with pytest.raises(NotImplementedError):
        ldsp.dsp_pmts(data, dict(cfg, wsg_weight=2))


def test_fir_grid_linearity_form_equals_per_point_form():
    """ldsp_fir_grid_run: the shared-pick-off form (one estimator-weighted pass z over the trace, then one dot product per grid
    point) against the per-point evaluation of every filter output (option fir_grid_per_point), on rt and ft scans."""
    cfg = ldsp.reference_test_icpc_config()
    wvfs = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(48, L, device="cuda", seed=88), 0.0, 16.0)
    ctx = ldsp.default_context()
    for fn in (lambda: ldsp.dsp_cusp_rt_optimization(wvfs, cfg, 500 * ldsp.us, ctx=ctx),
               lambda: ldsp.dsp_zac_ft_optimization(wvfs, cfg, 500 * ldsp.us, 8 * ldsp.us, ctx=ctx)):
        fast = fn().clone()
        ctx.set_option("fir_grid_per_point", 1)
        try:
            ref = fn().clone()
        finally:
            ctx.set_option("fir_grid_per_point", 0)
        torch.testing.assert_close(fast, ref, rtol=2e-5, atol=0.05)
