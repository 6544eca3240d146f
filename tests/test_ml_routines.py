"""QC classifier front end: the Haar feature kernel against the oracle, the RBF-SVM decision function
against libsvm itself (scikit-learn's SVC wraps the same libsvm the reference calls through LIBSVM.jl, src/ml.jl:6-22)."""
import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp


def _svc_model(X, y, gamma=0.5, C=1.0):
    from sklearn.svm import SVC
    clf = SVC(kernel="rbf", gamma=gamma, C=C, decision_function_shape="ovo").fit(X, y)
    # libsvm's own arrays (scikit-learn negates the public copies for two classes)
    model = dict(support_vectors=clf.support_vectors_, n_sv=clf.n_support_, dual_coef=clf._dual_coef_, rho=-clf._intercept_,
                 labels=clf.classes_, gamma=gamma)
    return clf, model


@pytest.mark.parametrize("k", [2, 3, 4])
def test_rbf_svm_predictor_equals_libsvm(k):
    rng = np.random.default_rng(5 + k)
    centers = rng.normal(size=(k, 16)) * 1.5
    y = rng.integers(0, k, 600)
    X = centers[y] + rng.normal(size=(600, 16))
    labels = np.array([0, 1, 3, 7])[:k][y]          # non-contiguous class labels, as the reference's dc_labels may be
    clf, model = _svc_model(X[:400], labels[:400], gamma=0.05)
    pred = ldsp.RbfSvmPredictor(**model, device="cpu")
    yp, dec = pred(torch.as_tensor(X[400:], dtype=torch.float32))
    ref_dec = clf.decision_function(X[400:]).reshape(200, -1)
    if k == 2:
        ref_dec = -ref_dec                           # scikit-learn's two-class sign flip (see _svc_model)
    np.testing.assert_allclose(dec.numpy(), ref_dec, atol=2e-4)
    sure = np.abs(ref_dec).min(axis=1) > 1e-3        # float32 vs float64 may flip a vote on the margin
    assert sure.sum() > 150
    assert np.array_equal(yp.numpy()[sure], clf.predict(X[400:])[sure])


@pytest.mark.gpu
@pytest.mark.parametrize("L,levels,with_bl", [(8192, 5, True), (8192, 2, False), (6251, 5, True), (4096, 2, True), (1000, 5, False)])
def test_qc_features_match_oracle(orc, L, levels, with_bl):
    n = 24
    cfg = ldsp.reference_test_icpc_config()
    wf = ldsp.synth.hpge_batch(n, 8192, device="cuda", seed=31)[:, :L].contiguous()
    wf[3] = wf[3, 0]                                 # constant trace: all-zero features after the shift, norm 0 -> 1
    w = ldsp.ArrayOfRDWaveforms(wf, 0.0, 16.0)
    feat, norm = ldsp.qc_features(w, levels, cfg if with_bl else None, return_norm=True)
    a = b = -1
    if with_bl:
        a, b = ldsp.config.window_index(cfg.bl_window.left, 0.0, 16.0), ldsp.config.window_index(cfg.bl_window.right, 0.0, 16.0)
    of, on = orc.qc_features(wf.cpu().numpy(), levels, a, b)
    assert feat.shape == of.shape
    np.testing.assert_allclose(norm.cpu().numpy(), on, rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(feat.cpu().numpy(), of, atol=3e-5 if with_bl else 2e-6)
    assert float(feat.abs().max()) <= 1.0 + 1e-6
    if with_bl:
        assert float(feat[3].abs().max()) < 1e-6 and float(norm[3]) <= 1.0


@pytest.mark.gpu
def test_dsp_icpc_with_qc_classifier(orc):
    """dsp_icpc(...; f_evaluate_qc) (src/dsp_icpc.jl:105-108): labels = libsvm's on the oracle's features."""
    n, L = 96, 8192
    cfg = ldsp.reference_test_icpc_config()
    wf = ldsp.synth.hpge_batch(n, L, device="cuda", seed=77)
    wf[::3] = wf[::3].flip(1)                        # a second population: falling traces
    a, b = ldsp.config.window_index(cfg.bl_window.left, 0.0, 16.0), ldsp.config.window_index(cfg.bl_window.right, 0.0, 16.0)
    of, _ = orc.qc_features(wf.cpu().numpy(), 5, a, b)
    y = np.where(np.arange(n) % 3 == 0, 1, 0)
    clf, model = _svc_model(of[::2], y[::2], gamma=0.5)
    f_evaluate_qc = ldsp.RbfSvmPredictor(**model)
    data = ldsp.Table(waveform=ldsp.ArrayOfRDWaveforms(wf, 0.0, 16.0), baseline=torch.zeros(n), timestamp=torch.arange(n),
                      eventnumber=torch.arange(1, n + 1), daqenergy=torch.zeros(n))
    res = ldsp.dsp_icpc(data, cfg, 500 * ldsp.us, {}, f_evaluate_qc=f_evaluate_qc)
    lab = res["qc_label"].cpu().numpy()
    assert lab.dtype == np.int64 and set(np.unique(lab)) == {0, 1}
    ref = clf.predict(of)
    sure = np.abs(clf.decision_function(of)) > 1e-3
    assert np.array_equal(lab[sure], ref[sure]) and sure.sum() > n * 0.9
    assert (lab == y).mean() > 0.95
    # the routines that take the classifier accept it too
    r2 = ldsp.dsp_qc_flt_optimization(data["waveform"], cfg, 500 * ldsp.us, f_evaluate_qc)
    assert set(np.unique(r2["qc_label"].cpu().numpy())) <= {0, 1}
