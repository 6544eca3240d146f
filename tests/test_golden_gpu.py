"""The HIP entry points against the committed fixtures of tests/golden/ (no oracle build needed):
the reference's own known-answer data, and the frozen oracle vectors for dsp_icpc / dsp_sipm."""
import os

import numpy as np
import pytest
import torch

import golden_cases
import legenddsp_jl_amd as ldsp
import parity

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _wv(c):
    x = golden_cases.input_of(c).astype(np.float32)[None]
    return ldsp.ArrayOfRDWaveforms(torch.from_numpy(x).cuda(), c["t0"], c["dt"])


def _hip_run(c):
    w, op = _wv(c), c["op"]
    row = lambda t: t.cpu().numpy()[0]
    if op == "haar":
        o = ldsp.HaarAveragingFilter(c["ds"])(w)
        assert o.dt == c["expect_dt"] and o.t_first == c["t0"]
        return row(o.signal)
    if op == "moving_window":
        return row(ldsp.MovingWindowFilter(c["length"])(w).signal)
    if op == "moving_window_multi":
        return row(ldsp.MovingWindowMultiFilter(c["length"])(w).signal)
    if op == "derivative":
        return row(ldsp.DerivativeFilter(c["gain"])(w).signal)
    if op == "get_wvf_maximum":
        return float(ldsp.get_wvf_maximum(w, c["start"], c["stop"])[0])
    if op == "intersect_maximum":
        r = ldsp.IntersectMaximum(mintot=c["mintot"], maxtot=c["maxtot"])(w, c["threshold"])
        out = {k: r[k][0].cpu().numpy() for k in ("x", "x_high", "x_tot", "max")}
        out["multiplicity"] = int(r["multiplicity"][0])
        return out
    if op == "multi_intersect":
        return row(ldsp.MultiIntersect(threshold_ratios=tuple(c["ratios"]), mintot=c["mintot"])(w))
    if op == "extremestats":
        r = ldsp.extremestats(w, c["start"], c["stop"]) if "start" in c else ldsp.extremestats(w)
        return {k: float(v[0]) for k, v in r.items()}
    if op == "thresholdstats_mad":
        lo = -float("inf") if c["lo"] is None else c["lo"]
        hi = float("inf") if c["hi"] is None else c["hi"]
        return float(ldsp.thresholdstats_mad(w, lo, hi)[0])
    raise AssertionError(op)


@pytest.mark.parametrize("case", golden_cases.load(), ids=golden_cases.case_id)
def test_reference_known_answers(case):
    golden_cases.check(case, _hip_run(case))


def test_icpc_against_frozen_oracle_vectors():
    g = np.load(os.path.join(GOLD, "icpc_oracle_vectors.npz"))
    wf = torch.from_numpy(g["wf"]).cuda()
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, wf.shape[1], 0.0, 16.0)
    tab = ldsp.icpc_run(wf, p)
    gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
    ora = {str(c): g["table"][:, j] for j, c in enumerate(g["columns"])}
    lines, worst = parity.compare(gpu, ora)
    assert worst == 0.0, "\n".join(lines)      # 9 traces: no threshold flips expected, every column of every trace within tolerance
    pz = ldsp.icpc_pz_trap_run(wf, p).cpu().numpy()
    np.testing.assert_allclose(pz[0], g["pz_blmean"], atol=2e-3)
    np.testing.assert_allclose(pz[1], g["pz_e10410"], rtol=2e-5, atol=0.05)


def test_sipm_against_frozen_oracle_vectors():
    from test_sipm_gpu import _compare
    s = np.load(os.path.join(GOLD, "sipm_oracle_vectors.npz"))
    wf = torch.from_numpy(s["wf"]).cuda()
    p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, wf.shape[1], 0.0, 16.0)
    sc, trig = ldsp.sipm_run(wf, p)
    ora = {c: s["col__" + c] for c in ldsp._abi.SIPM_SCALAR_COLS}
    for grp in ldsp._abi.SIPM_TRIG_GROUPS:
        ora[grp] = {f: s[f"trig__{grp}__{f}"] for f in ("count", "x", "x_high", "x_tot", "max")}
    assert _compare(sc, trig, ora, wf.shape[0], wf, p, None) == (0, 0)   # (no oracle: flat 0.01 ns on every position, counts equal)
