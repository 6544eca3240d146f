"""Parity of the fused HIP dsp_icpc kernel with the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp
import parity

pytestmark = pytest.mark.gpu

L = 8192


@pytest.fixture(scope="module")
def params():
    return ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)


def _run(wf, params, direct=0, generic=0):
    """direct: CUSP/ZAC as direct-form FIR (comparator of the closed form); generic: icpc_kernel instead of the lean kernel
    (icpc_lean.hip) where both apply"""
    ctx = ldsp.default_context()
    ctx.set_option("cusp_direct", direct)
    ctx.set_option("icpc_generic", generic)
    try:
        tab = ldsp.icpc_run(wf, params, ctx)
        torch.cuda.synchronize()
    finally:
        ctx.set_option("cusp_direct", 0)
        ctx.set_option("icpc_generic", 0)
    return {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}


def test_native_library_loaded():
    import ctypes
    assert isinstance(ldsp._lib.lib(), ctypes.CDLL)
    with open("/proc/self/maps") as f:
        assert "libldsp_hip.so" in f.read()


@pytest.mark.parametrize("direct,generic", [(0, 0), (0, 1), (1, 0)])
def test_icpc_matches_oracle_seeded_batch(orc, params, direct, generic):
    """(0, 0): the lean kernel (the production path at this geometry); (0, 1): icpc_kernel on the same batch; (1, 0): direct-form
    CUSP/ZAC (two launches of the generic kernels)"""
    n = 512 if direct == 0 else 128
    wf = ldsp.synth.hpge_batch(n, L, device="cuda")
    gpu = _run(wf, params, direct, generic)
    ora = orc.dsp_icpc(wf.cpu().numpy(), params, nthreads=16)
    lines, worst = parity.compare(gpu, ora)
    print("\n".join(lines))
    assert worst <= parity.FLIP_FRAC, "\n".join(lines)


def test_synthetic_batch_is_the_same_on_host_and_device():
    """the synthetic input is counter-based so that the bench batch can be regenerated anywhere — any rows of it, on the
    host: 64 traces from the middle of the 1 M-trace batch, generated on the GPU and on the CPU, are the same numbers (the integer
    hash is exact on both; the Gaussian transform is rounded to float32 from double, so at most a last-place difference in isolated
    samples)."""
    a = ldsp.synth.hpge_batch(64, L, device="cuda", first_trace=500_000).cpu()
    b = ldsp.synth.hpge_batch(64, L, device="cpu", first_trace=500_000)
    diff = a != b
    assert int(diff.sum()) <= 8, int(diff.sum())                      # of 524 288 samples
    assert float((a - b).abs().max()) <= 2.5e-4                        # one ulp at 1e3 .. 2e4 ADC counts
    s1 = ldsp.synth.sipm_batch(16, 16384, device="cuda", first_trace=300_000).cpu()
    s2 = ldsp.synth.sipm_batch(16, 16384, device="cpu", first_trace=300_000)
    assert float((s1 - s2).abs().max()) <= 2e-6 and int((s1 != s2).sum()) <= 64


def test_lean_and_generic_kernels_agree(params):
    """The two implementations of the chain (icpc_lean3_kernel: pivoted sums, one exchange; icpc_kernel: the round-1 form) against
    EACH OTHER on the seeded batch, tighter than either is held to the oracle: a regression in the run-count or mask logic of one
    of them shows here even where the oracle budget (inTrace_n within 3, parity.py) would let it pass.  Integer columns equal on
    every row but threshold-decision flips (each kernel rounds its own sigma): at most 1 % of rows, and then by one count."""
    n = 512
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=7)
    lean, gen = _run(wf, params, 0, 0), _run(wf, params, 0, 1)
    for c in parity.INT_COLS:
        a, b = lean[c].astype(np.int64), gen[c].astype(np.int64)
        diff = a != b
        assert diff.sum() <= n // 100, (c, int(diff.sum()))
        assert np.abs(a - b).max() <= 1, (c, a[diff][:8], b[diff][:8])
    for c, tol in (("blmean", 2e-4), ("e_max", 2e-3), ("e_10410", 0.02), ("e_trap", 0.02), ("e_cusp", 0.05), ("e_zac", 0.05), ("t50", 2e-5), ("t0", 2e-5)):
        d = np.abs(lean[c].astype(np.float64) - gen[c].astype(np.float64))
        assert (d > tol).sum() <= n // 100, (c, float(np.nanmax(d)))


def test_float32_envelope(orc, params):
    """The Float32-typed restatement (oracle/ldsp_oracle.c -DORC_F32: what the reference computes for Float32 input) against the
    Float64 one on the seeded batch: printed per column next to the HIP path's error (tests/parity.py, module text).  The HIP
    path must lie INSIDE that envelope wherever the envelope is wider than the column's own floor — it does by one to three orders
    of magnitude (no column of a statistic, energy, drift or current may come closer than a third of it)."""
    n = 256
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=101)
    host = wf.cpu().numpy()
    gpu = _run(wf, params)
    ora = orc.dsp_icpc(host, params, nthreads=16)
    env = orc.dsp_icpc(host, params, nthreads=16, f32=True)
    lines, worst = parity.compare(gpu, ora, wf=host, params=params, orc=orc, env=env)
    print("\n".join(lines))
    assert worst <= parity.FLIP_FRAC, "\n".join(lines)
    for c in ("tailmean", "tailsigma", "tailoffset", "blsigma", "e_10410", "e_535", "e_313", "e_trap", "e_cusp", "e_trap_max", "e_cusp_max", "qdrift", "lq"):
        e_hip = np.nanmax(np.abs(np.asarray(gpu[c], dtype=np.float64) - ora[c]))
        e_f32 = np.nanmax(np.abs(env[c] - ora[c]))
        assert e_hip <= e_f32 / 3, (c, e_hip, e_f32)


def test_reference_fake_waveform_properties(orc, params):
    """The smoke properties the reference asserts (test/test_dsp_icpc.jl:189-199) plus the
    analytic consequences of its noiseless fixture; and the whole table against the oracle — the pile-up columns
    too where the Float32 and the Float64 restatement agree with each other on this noise-free trace (tests/parity.py)."""
    wf = ldsp.synth.reference_hpge_waveform().float()[None].repeat(3, 1).cuda()
    g = _run(wf, params)
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, params, nthreads=2)
    env = orc.dsp_icpc(host, params, nthreads=2, f32=True)
    lines, rows = parity.compare_rows(g, ora, wf=host, params=params, orc=orc, env=env)
    print("\n".join(l for l in lines if l.startswith("inTrace")))
    assert rows == 0, "\n".join(lines)
    assert np.all(g["t0"] < g["t50"]) and np.all(g["t50"] < g["t90"]) and np.all(g["drift_time"] >= 0)
    for c in ("e_10410", "e_313", "e_trap", "e_cusp", "e_zac"):
        assert np.all(np.isfinite(g[c]))
    np.testing.assert_allclose(g["blmean"], 1000.0, rtol=1e-6)
    np.testing.assert_allclose(g["e_max"], 10000.0, rtol=1e-6)
    np.testing.assert_allclose(g["tail_tau"], 500000.0, rtol=1e-4)
    np.testing.assert_allclose(g["e_10410"], 10020.16, rtol=1e-5)
    assert np.all(g["n_sat_low"] == 0) and np.all(g["inTrace_n"] >= 0)
    # identical inputs -> bitwise identical rows (no atomics in the design)
    for c, v in g.items():
        assert np.array_equal(v[0:1].repeat(3, 0), v, equal_nan=True), c


def test_edge_cases_saturation_and_flat(orc, params):
    wf = ldsp.synth.hpge_batch(8, L, device="cuda").clone()
    wf[0, 100:140] = 0.0            # 40 consecutive samples saturated low
    wf[0, 500:503] = 0.0
    wf[1, 4000:4100] = 65520.0      # saturated high
    wf[2, :] = 1000.0               # flat trace: no crossings anywhere (NaN -> 0 paths), tailstats <= 0 branch
    wf[3, :] = wf[3, :].flip(0)     # pulse reversed: falling edge
    gpu = _run(wf, params)
    ora = orc.dsp_icpc(wf.cpu().numpy(), params, nthreads=4)
    assert gpu["n_sat_low"][0] == 43 and gpu["n_sat_low_cons"][0] == 40
    assert gpu["n_sat_high"][1] == 100 and gpu["n_sat_high_cons"][1] == 100
    assert gpu["t0"][2] == 0 and gpu["t50"][2] == 0 and gpu["tail_tau"][2] == 0
    for c in ldsp._abi.ICPC_I32_COLS:
        np.testing.assert_array_equal(gpu[c], ora[c].astype(np.int64), err_msg=c)
    lines, worst = parity.compare({k: v[4:] for k, v in gpu.items()}, {k: v[4:] for k, v in ora.items()})
    assert worst == 0, "\n".join(lines)


def test_other_lengths_and_filter_parameters(orc):
    """Config 1 plumbing case (L = 4096 @ 32 ns) and non-default pars_filter (CUSP != ZAC geometry)."""
    cfg = ldsp.reference_test_icpc_config()
    p1 = ldsp.lower_icpc(ldsp.plumbing_icpc_config_4096(), 500 * ldsp.us, {}, 4096, 0.0, 32.0)
    wf = ldsp.synth.hpge_batch(64, 4096, device="cuda")
    gpu = _run(wf, p1)
    ora = orc.dsp_icpc(wf.cpu().numpy(), p1, nthreads=8)
    lines, rows = parity.compare_rows(gpu, ora)
    assert rows <= 1, "\n".join(lines)
    pf = {"trap": {"rt": 8 * ldsp.us, "ft": 3 * ldsp.us}, "cusp": {"rt": 4 * ldsp.us, "ft": 2 * ldsp.us},
          "zac": {"rt": 6 * ldsp.us, "ft": 1.5 * ldsp.us}, "sg": {"wl": 180 * ldsp.ns}}
    p2 = ldsp.lower_icpc(cfg, 450 * ldsp.us, pf, L, 0.0, 16.0)
    wf = ldsp.synth.hpge_batch(64, L, device="cuda", seed=7)
    gpu = _run(wf, p2)
    ora = orc.dsp_icpc(wf.cpu().numpy(), p2, nthreads=8)
    lines, rows = parity.compare_rows(gpu, ora)
    assert rows <= 1, "\n".join(lines)


def test_pz_trap_subchain(orc, params):
    wf = ldsp.synth.hpge_batch(256, L, device="cuda", seed=3)
    out = ldsp.icpc_pz_trap_run(wf, params).cpu().numpy()
    ora = orc.icpc_pz_trap(wf.cpu().numpy(), params)
    np.testing.assert_allclose(out[0], ora["blmean"], rtol=1e-6)
    np.testing.assert_allclose(out[1], ora["e_10410"], rtol=2e-6)


def test_full_size_properties(params):
    """Size-independent properties at a large batch: linearity of the chain in the
    amplitude (energies scale, times do not) and shift invariance in the baseline."""
    n = 4096
    base = ldsp.synth.hpge_batch(n, L, device="cuda", noise=0.0, seed=11)
    g1 = _run(base, params)
    bl = base[:, :2000].mean(dim=1, keepdim=True)
    g2 = _run((base - bl) * 2.0 + bl + 37.0, params)
    # (qdrift is not amplitude-linear: it hangs on t0, picked off at a FIXED 4-ADC threshold)
    for c in ("e_10410", "e_535", "e_313", "e_trap", "e_cusp", "e_zac", "e_max", "lq", "a_sg"):
        np.testing.assert_allclose(g2[c], 2.0 * g1[c], rtol=2e-4, atol=2e-2 * max(1.0, np.abs(g1[c]).max() * 1e-4), err_msg=c)
    for c in ("t10", "t50", "t90", "t99"):
        np.testing.assert_allclose(g2[c], g1[c], atol=2e-3, err_msg=c)
    np.testing.assert_allclose(g2["blmean"], g1["blmean"] + 37.0 + (g1["blmean"] - g1["blmean"]), rtol=1e-5)


@pytest.mark.parametrize("Lx", [8000, 8190, 7001])
def test_trace_lengths_that_do_not_fill_the_tile(orc, Lx):
    """L < 16*NT (bounds-tested variants of the kernels), a multiple of 4, L % 4 == 2 and odd L (rows not 16-byte aligned)."""
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, Lx, 0.0, 16.0)
    wf = ldsp.synth.hpge_batch(48, 8192, device="cuda", seed=13)[:, :Lx].contiguous()
    gpu = _run(wf, p)
    ora = orc.dsp_icpc(wf.cpu().numpy(), p, nthreads=8)
    lines, rows = parity.compare_rows(gpu, ora)
    assert rows <= 1, "\n".join(lines)


def test_two_launch_form_equals_fused(params, orc):
    """`two_kernel` = 1 (icpc_kernel + icpc_cz_kernel, the fall-back form) gives the fused generic launch's table to the
    last bits, and the lean kernel's table within the oracle budgets (a different summation order)."""
    wf = ldsp.synth.hpge_batch(96, L, device="cuda", seed=17)
    fused = _run(wf, params, generic=1)
    lean = _run(wf, params)
    ctx = ldsp.default_context()
    ctx.set_option("two_kernel", 1)
    try:
        two = _run(wf, params)
    finally:
        ctx.set_option("two_kernel", 0)
    fused64 = {k: np.asarray(v, dtype=np.float64) for k, v in fused.items()}
    for c in ldsp._abi.ICPC_COLS:
        if c in parity.TIME_MAX:      # an arg-max may sit on a tie that the two summation orders break differently
            bad, _ = parity.bad_mask(c, two, fused64)
            assert bad.sum() == 0, c
        else:
            np.testing.assert_allclose(two[c], fused[c], rtol=3e-6, atol=1e-3, equal_nan=True, err_msg=c)
    lines, worst = parity.compare(two, {k: np.asarray(v, dtype=np.float64) for k, v in lean.items()})
    assert worst <= parity.FLIP_FRAC, "\n".join(lines)


def test_long_traces_16384(orc):
    """L = 16384 (NT = 1024): one trace per CU, the CUSP/ZAC stage runs as the second launch (LDS budget)."""
    Lx = 16384
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, Lx, 0.0, 16.0)
    wf = ldsp.synth.hpge_batch(32, Lx, device="cuda", seed=19)
    gpu = _run(wf, p)
    ora = orc.dsp_icpc(wf.cpu().numpy(), p, nthreads=8)
    lines, rows = parity.compare_rows(gpu, ora)
    assert rows <= 1, "\n".join(lines)


def test_threshold_crossings_with_spikes_before_the_pulse(orc, params):
    """get_threshold (src/dsp_routines.jl:33-42) = Intersect(mintot): the FIRST up-crossing that holds for tx_mintot samples.
    The kernel confirms the first sample at or above each threshold and falls back to the general bit-mask scan when the
    confirmation fails: single-sample spikes in the baseline (rejected by mintot = 2 samples), a spike pair (accepted: it IS the
    first crossing), a trace that starts above the thresholds (initial run: no crossing there), a flat trace (e_max = 0)."""
    n = 64
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=77)
    amp = (wf[:, 6000:6100].mean(dim=1) - wf[:, :2000].mean(dim=1))      # pulse height above baseline
    for i in range(0, 16):           # one-sample spike above the q-th threshold, well before the rise: must be skipped
        frac = (0.15, 0.55, 0.85, 0.95, 1.2)[i % 5]
        wf[i, 500 + 37 * i] += frac * amp[i]
    for i in range(16, 24):          # two-sample spike: a genuine first crossing of the lower thresholds
        wf[i, 700 + 11 * i] += 0.6 * amp[i]
        wf[i, 701 + 11 * i] += 0.6 * amp[i]
    for i in range(24, 28):          # spike in sample 0 and 1: a run that starts the trace is not a crossing
        wf[i, 0:2] += 0.7 * amp[i]
    for i in range(28, 32):          # the trace starts high and comes down: thresholds crossed downwards first
        wf[i, :300] += 1.5 * amp[i]
    wf[32] = 1000.0                  # flat: e_max = 0
    wf[33] = 1000.0 + 0.25 * torch.sin(torch.arange(L, device="cuda") / 50.0)     # tiny ripple: thresholds within the noise
    ora = orc.dsp_icpc(wf.cpu().numpy(), params, nthreads=16)
    smooth = np.ones(n, dtype=bool)
    smooth[24:34] = False    # steps of the full pulse height next to the windows: qdrift / lq there move by hundreds per 0.01 sample of t0
    for generic in (0, 1):   # the lean kernel and icpc_kernel share the confirmation + fall-back scheme
        gpu = _run(wf, params, generic=generic)
        for c in ("t10", "t50", "t80", "t90", "t99", "t0", "t0_inv", "drift_time", "qdrift", "lq", "e_trap", "e_cusp", "e_zac"):
            a, b = gpu[c].astype(np.float64), ora[c]
            tol = 5e-4 if c.startswith("t") else (parity.ATOL[c] + parity.RTOL * np.abs(b))
            if c == "drift_time":
                tol = 0.6
            bad = ~((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b)))
            if c in ("qdrift", "lq"):
                bad &= smooth
            assert bad.sum() == 0, (generic, c, np.nonzero(bad)[0], a[bad], b[bad])
        # the spikes did what they were meant to: traces 0..15 keep their crossing at the pulse, 16..23 moved to the spike pair
        assert np.all(gpu["t10"][:16] > 40.0) and np.all(gpu["t10"][16:24] < 20.0)

@pytest.mark.parametrize("length,kernel,sep", [(8000, "lean3::icpc_lean3_kernel", False), (7300, "lean3::icpc_lean3_kernel", False),
                                                (4400, "lean3::icpc_lean3_kernel", False), (7600, "lean3::icpc_lean3_kernel", True),
                                                (8190, "lean3::icpc_lean3_kernel", False), (7301, "lean3::icpc_lean3_kernel", False),
                                                (4403, "lean3::icpc_lean3_kernel", False), (7602, "lean3::icpc_lean3_kernel", True),
                                                (4096 + 1, "lean3::icpc_lean3_kernel", False), (8191, "lean3::icpc_lean3_kernel", True)])
def test_traces_shorter_than_the_tile(orc, length, kernel, sep):
    """A trace that does not fill the tile (16 x threads samples) still runs the fused lean kernel (a lane whose quad lies beyond
    the trace keeps a copy of its own row-0 quad and every output range is bounded by the length) — also when its length is no
    multiple of four samples (round 4: the rows are then 4-byte aligned and the one quad that holds the end of a trace is read
    sample by sample): lengths % 4 = 0, 1, 2, 3, one sample more than half the tile, one less than the tile.  Against the oracle on
    traces whose pulse, tail and windows reach to the very end of the trace, and against the generic kernel."""
    sc = length / 8192.0
    dt = 16.0
    us = ldsp.us
    import dataclasses
    cfg = dataclasses.replace(ldsp.reference_test_icpc_config(),
                              bl_window=ldsp.ClosedInterval(0.0, 39.0 * us * sc), tail_window=ldsp.ClosedInterval(70.0 * us * sc, (length - 1) * dt),
                              current_window=ldsp.ClosedInterval(43.0 * us * sc, 62.0 * us * sc),
                              flt_length_cusp=38.0 * us * sc, flt_length_zac=38.0 * us * sc)
    pf = {"cusp": {"rt": 4.0 * us * sc, "ft": 1.5 * us * sc}, "zac": {"rt": 5.5 * us * sc, "ft": 2.0 * us * sc}} if sep else {}   # sep: two passes of the closed-form stage
    p = ldsp.lower_icpc(cfg, 500 * us, pf, length, 0.0, dt)
    n = 128
    wf = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=61)
    idx = (torch.arange(length, device="cuda", dtype=torch.float32) / sc).long().clamp(max=8191)
    wf = wf[:, idx].contiguous()                     # the 8192-sample shapes resampled onto `length` samples
    wf[:4, -3:] += 500.0                             # a step in the very last samples: nothing beyond the trace may answer it
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16)
    ctx = ldsp.default_context()
    gpu = _run(wf, p)
    assert ctx.last_kernel_name() == kernel
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    assert worst <= 2 / n, "\n".join(l for l in lines if f"bad=0/{n}" not in l)
    if kernel.startswith("lean3"):
        gen = _run(wf, p, generic=1)
        assert ctx.last_kernel_name() == "icpc_kernel"
        for c in parity.INT_COLS:
            assert (np.abs(gpu[c].astype(np.int64) - gen[c].astype(np.int64)) > 0).sum() <= 2, c
        # the saturation counters of a trace clipped at the rail in its last samples: the lanes beyond the trace count nothing
        sat = wf.clone(); sat[:, -5:] = p.sat_high
        a, b = _run(sat, p), _run(sat, p, generic=1)
        for c in ("n_sat_high", "n_sat_high_cons", "n_sat_low", "n_sat_low_cons"):
            assert np.array_equal(a[c], b[c]) and (c != "n_sat_high" or np.all(a[c] >= 5)), c


def test_context_changes_stream_and_can_leave_a_destroyed_one(params):
    """ldsp_ctx_set_stream (include/ldsp.h, lifetime rule): run on stream A, switch to B (ordered behind A by an event), destroy A's
    successor while the context still points at it, and switch again — the context must adopt the new stream (it waits for the
    event recorded at the end of its last run and never touches the old stream) instead of failing on the dead one for ever; the
    tables are the same every time."""
    wf = ldsp.synth.hpge_batch(64, L, device="cuda", seed=5)
    torch.cuda.synchronize()                         # (the batch is complete before another stream reads it)
    ctx = ldsp.Context(0, use_torch_stream=False)   # (the context launches where set_stream says, not on torch's current stream)
    t = ldsp.icpc_run(wf, params, ctx)
    ctx.synchronize()                                # (the context's own stream: torch does not see it)
    ref = t.clone()
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    for st in (a, b):
        st.wait_stream(torch.cuda.current_stream())
        ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            t = ldsp.icpc_run(wf, params, ctx)
        st.synchronize()
        assert torch.equal(t.view(torch.int32), ref.view(torch.int32))
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
    ctx.set_stream(raw.value)
    t = ldsp.icpc_run(wf, params, ctx)
    ctx.synchronize()
    assert torch.equal(t.view(torch.int32), ref.view(torch.int32))
    assert hip.hipStreamDestroy(raw) == 0            # the context still points at it
    ctx.set_stream(b.cuda_stream)                    # must succeed (or report once) and adopt b
    with torch.cuda.stream(b):
        t = ldsp.icpc_run(wf, params, ctx)
    b.synchronize()
    assert torch.equal(t.view(torch.int32), ref.view(torch.int32))
    ctx.use_own_stream()
    t = ldsp.icpc_run(wf, params, ctx)
    ctx.synchronize()
    assert torch.equal(t.view(torch.int32), ref.view(torch.int32))


def _scaled_config(length, dt, degree2):
    """The reference test configuration with every window scaled onto a trace of `length` samples at `dt` ns (the fixed filters keep
    their times; at dt >= 32 ns the 60 ns Savitzky-Golay window and the 100 ns estimator cannot carry a cubic: degree 2, as
    plumbing_icpc_config_4096 does)."""
    import dataclasses
    us = ldsp.us
    sc = length * dt / (8192 * 16.0)
    cfg = ldsp.plumbing_icpc_config_4096() if degree2 else ldsp.reference_test_icpc_config()
    return dataclasses.replace(cfg, bl_window=ldsp.ClosedInterval(0.0, 39.0 * us * sc),
                               tail_window=ldsp.ClosedInterval(70.0 * us * sc, min(110.0 * us * sc, (length - 1) * dt)),
                               current_window=ldsp.ClosedInterval(43.0 * us * sc, 62.0 * us * sc),
                               enc_pickoff_trap=40.0 * us * sc, enc_pickoff_zac=41.0 * us * sc, enc_pickoff_cusp=41.0 * us * sc,
                               flt_length_cusp=38.0 * us * sc, flt_length_zac=38.0 * us * sc), sc


@pytest.mark.parametrize("length,dt,u16,sep", [(2048, 16.0, False, False), (1024, 32.0, False, False),      # full tiles of 128 / 64 threads
                                                 (4096, 32.0, False, True),                                   # 256 threads, CUSP and ZAC separate
                                                 (3000, 16.0, False, False), (1500, 16.0, True, False),       # short tiles of 256 / 128 threads (uint16 input)
                                                 (900, 32.0, False, True), (1800, 16.0, True, True),          # short tiles of 64 / 128 threads, separate CUSP / ZAC
                                                 (3001, 16.0, True, False), (1503, 16.0, False, False), (1798, 16.0, True, True)])   # lengths that are no multiple of four samples (uint16 rows 2-byte aligned)
def test_small_tiles_run_the_lean_kernel(orc, length, dt, u16, sep):
    """Every instantiation the launcher admits (ldsp_api.hip: icpc_lean_applies) is compared with the oracle by name: tiles of 64, 128
    and 256 threads (L = 1024, 2048, 4096), full and shorter than the tile, float32 and uint16 input, shared and separately optimised
    CUSP / ZAC — and with the generic kernel on the same batch.  (Round 3 tested the 512-thread tile only; the round-2 advisor had found
    exactly this class of untested instantiation broken in the previous kernel.)"""
    us = ldsp.us
    cfg, sc = _scaled_config(length, dt, degree2=dt >= 32.0)
    pf = {"cusp": {"rt": 4.0 * us * sc, "ft": 1.5 * us * sc}, "zac": {"rt": 5.5 * us * sc, "ft": 2.0 * us * sc}} if sep else {}
    pf["trap"] = {"rt": 5.0 * us * sc, "ft": 2.5 * us * sc}
    p = ldsp.lower_icpc(cfg, 500 * us, pf, length, 0.0, dt)
    n = 128
    wf = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=71)
    idx = (torch.arange(length, device="cuda", dtype=torch.float32) * (8192.0 / length)).long().clamp(max=8191)
    wf = wf[:, idx].contiguous()                     # the 8192-sample shapes resampled onto `length` samples
    if u16:
        wf = wf.round().clamp(0, 65535)
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16)
    ctx = ldsp.default_context()
    gpu = _run(wf.to(torch.uint16) if u16 else wf, p)
    assert ctx.last_kernel_name() == "lean3::icpc_lean3_kernel"
    again = _run(wf.to(torch.uint16) if u16 else wf, p)     # a second launch gives the same bits (no race between the tile's waves)
    for c in gpu:
        assert np.array_equal(gpu[c].view(np.int32), again[c].view(np.int32)), c
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    detail = []
    if worst > 2 / n:   # (which rows, and what both sides hold there)
        for c in ldsp._abi.ICPC_COLS:
            bad, _ = parity.bad_mask(c, gpu, ora, host, p, orc)
            if bad.any():
                r = np.nonzero(bad)[0][:6]
                detail.append(f"{c}: rows {r.tolist()} gpu {np.asarray(gpu[c])[r].tolist()} oracle {np.asarray(ora[c])[r].tolist()}")
    assert worst <= 2 / n, "\n".join([l for l in lines if f"bad=0/{n}" not in l] + detail)
    gen = _run(wf, p, generic=1)
    assert ctx.last_kernel_name() == "icpc_kernel"
    for c in parity.INT_COLS:
        assert (np.abs(gpu[c].astype(np.int64) - gen[c].astype(np.int64)) > 0).sum() <= 2, c


def test_separately_optimised_cusp_and_zac_run_in_the_lean_launch(orc):
    """pars_filter with different rise / flat-top times for CUSP and ZAC (what an optimisation campaign produces): the lean
    kernel evaluates the two filters in two passes of its closed-form stage inside the SAME launch (y kept in registers for
    the second one); the generic path needs three launches.  Both against the oracle, and the launch really is the lean kernel."""
    us = ldsp.us
    pf = {"cusp": {"rt": 4.0 * us, "ft": 1.5 * us}, "zac": {"rt": 5.5 * us, "ft": 2.0 * us}}
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, pf, L, 0.0, 16.0)
    assert p.cusp.length != p.zac.length or p.cusp.flat != p.zac.flat
    wf = ldsp.synth.hpge_batch(256, L, device="cuda", seed=31)
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16)
    ctx = ldsp.default_context()
    for generic, name in ((0, "lean3::icpc_lean3_kernel"), (1, "icpc_kernel")):
        gpu = _run(wf, p, generic=generic)
        assert ctx.last_kernel_name() == name
        lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
        assert worst <= parity.FLIP_FRAC, "\n".join(lines)


@pytest.mark.parametrize("wl_ns,taps,length,sep,kernel", [(240.0, 15, 8192, False, "lean3::icpc_lean3_kernel"), (300.0, 19, 8192, True, "lean3::icpc_lean3_kernel"),
                                                          (350.0, 23, 8000, False, "lean3::icpc_lean3_kernel"), (400.0, 25, 8190, False, "lean3::icpc_lean3_kernel"),
                                                          (430.0, 27, 8192, False, "icpc_kernel")])
def test_optimised_savitzky_golay_windows_run_the_lean_kernel(orc, wl_ns, taps, length, sep, kernel):
    """pars_filter.sg.wl from the reference's scan grid (30 ... 350 ns, test/test_dsp_icpc.jl:134-138; 22 samples -> 23 taps at 16 ns):
    windows of up to 25 taps run the fused lean kernel since round 4 (the main filter streams through the halo quads instead of keeping
    its window in registers; until round 3 more than 13 taps meant the generic kernel at half the rate), full and short tiles, shared
    and separate CUSP / ZAC; 27 taps still run icpc_kernel.  Against the oracle, and the lean kernel against the generic one."""
    us = ldsp.us
    pf = {"sg": {"wl": wl_ns * ldsp.ns}}
    if sep:
        pf.update({"cusp": {"rt": 4.0 * us, "ft": 1.5 * us}, "zac": {"rt": 5.5 * us, "ft": 2.0 * us}})
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, pf, length, 0.0, 16.0)
    assert p.sg_npts[0] == taps
    n = 128
    wf = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=83)[:, :length].contiguous()
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16)
    ctx = ldsp.default_context()
    gpu = _run(wf, p)
    assert ctx.last_kernel_name() == kernel
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    assert worst <= 2 / n, "\n".join(l for l in lines if f"bad=0/{n}" not in l)
    if kernel.startswith("lean3"):
        gen = _run(wf, p, generic=1)
        assert ctx.last_kernel_name() == "icpc_kernel"
        for c in parity.INT_COLS:
            assert (np.abs(gpu[c].astype(np.int64) - gen[c].astype(np.int64)) > 0).sum() <= 2, c
        for c in ("a_sg", "t50_current"):
            np.testing.assert_allclose(gpu[c], gen[c], rtol=3e-5, atol=1e-3, err_msg=c)


@pytest.mark.parametrize("generic,two_kernel", [(0, 0), (1, 0), (0, 1)])
def test_uint16_adc_counts_are_converted_by_the_kernel(params, generic, two_kernel):
    """ldsp_icpc_opts.in_u16: the traces as uint16 ADC counts (what production waveforms are) give the table of the same
    values passed as float32, bit for bit — the conversion happens in the kernel's load (lean, generic and two-launch form),
    not in a separate cast pass."""
    wf = ldsp.synth.hpge_batch(128, L, device="cuda", seed=41).round().clamp(0, 65535)
    wf16 = wf.to(torch.uint16)
    assert torch.equal(wf16.to(torch.float32), wf)
    ctx = ldsp.default_context()
    ctx.set_option("two_kernel", two_kernel)
    try:
        a, b = _run(wf, params, generic=generic), _run(wf16, params, generic=generic)
    finally:
        ctx.set_option("two_kernel", 0)
    for c in ldsp._abi.ICPC_COLS:
        assert np.array_equal(a[c], b[c], equal_nan=True), c


@pytest.mark.parametrize("length,dt", [(8192, 16.0), (8000, 16.0), (4096, 32.0)])
def test_pz_trap_subchain_on_uint16_adc_counts(length, dt):
    """ldsp_icpc_pz_trap_run_u16: uint16 ADC counts give blmean / e_10410 of the same values passed as float32, bit for bit
    (lean kernel at 8192 and 4096 samples, generic kernel at 8000), and they are the fused chain's columns: blmean bit for bit (8000: to the last bit, two kernels);
    e_10410 to the last bits (round 3: the fused chain takes T = prefix sum of the pole-zero output from ONE exchange of partial
    sums, config 2's kernel from two — the same sums in another order, each rounded once at the magnitude of T, 1e7..1e8)."""
    cfg = ldsp.plumbing_icpc_config_4096() if length == 4096 else ldsp.reference_test_icpc_config()
    p = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, length, 0.0, dt)
    wf = ldsp.synth.hpge_batch(192, 8192, device="cuda", seed=43).round().clamp(0, 65535)
    wf = (wf[:, ::2] if dt != 16.0 else wf[:, :length]).contiguous()
    wf16 = wf.to(torch.uint16)
    assert torch.equal(wf16.to(torch.float32), wf)
    a = ldsp.icpc_pz_trap_run(wf, p).cpu().numpy()
    b = ldsp.icpc_pz_trap_run(wf16, p).cpu().numpy()
    assert np.array_equal(a, b)
    full = ldsp.table_columns(ldsp.icpc_run(wf16, p))
    if length in (8192, 4096):      # the same statement in both lean kernels
        assert np.array_equal(b[0], full["blmean"].cpu().numpy())
    else:                           # 8000 samples: config 2 runs pz_trap_kernel, the fused chain icpc_lean3_kernel (shorter-than-tile traces)
        np.testing.assert_allclose(b[0], full["blmean"].cpu().numpy(), rtol=2.5e-7)
    e_full = full["e_10410"].cpu().numpy().astype(np.float64)
    assert np.all(np.abs(b[1] - e_full) <= 0.02 + 3e-6 * np.abs(e_full)), np.abs(b[1] - e_full).max()


def test_context_keeps_two_parameter_blocks():
    """The context holds the previous parameter block beside the current one (ldsp_ctx.hpp): a caller alternating between two blocks
    — dsp_icpc_compressed, presummed and windowed traces — gets the same tables as from fresh contexts, also after a third block has
    taken a slot and after the slots have changed places several times."""
    cfg = ldsp.reference_test_icpc_config()
    wf = ldsp.synth.hpge_batch(64, L, device="cuda", seed=9)
    blocks = [ldsp.lower_icpc(cfg, tau * ldsp.us, {}, L, 0.0, 16.0) for tau in (500, 300, 420)]
    fresh = [ldsp.icpc_run(wf, b, ldsp.Context(0)).clone() for b in blocks]
    ctx = ldsp.Context(0)
    for i in (0, 1, 0, 1, 2, 1, 0, 2, 2, 0):
        t = ldsp.icpc_run(wf, blocks[i], ctx)
        assert torch.equal(torch.nan_to_num(t, nan=-1.0), torch.nan_to_num(fresh[i], nan=-1.0)), i
    assert not torch.equal(torch.nan_to_num(fresh[0], nan=-1.0), torch.nan_to_num(fresh[1], nan=-1.0))
