"""QC classifier front end: `get_qc_classifier` / `get_qc_classifier_compressed`
(reference src/dsp_ml_routines.jl:9-70) and an RBF-SVM decision function to stand where the reference passes
`Base.Fix1(svmpredict, model)` (src/ml.jl:6-22, LIBSVM.jl — external).

The front end is one HIP kernel (`ldsp_qc_features`): optional baseline subtraction, HaarAveragingFilter(2) five (two)
times, division by max(|min|, |max|).  Its output is the feature matrix the reference hands to `f_evaluate_qc`,
laid out [n][Lout] (the memory of the reference's `flatview(VectorOfSimilarArrays(...))`, one trace per column there).
`f_evaluate_qc(features) -> (y_pred, decision_values)` is any callable with LIBSVM.svmpredict's return convention.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .config import DSPConfig, window_index, WindowError
from .routines import ArrayOfRDWaveforms


def qc_features(wvfs: ArrayOfRDWaveforms, levels: int, config: DSPConfig = None, ctx: _lib.Context = None, return_norm=False):
    """Normalised Haar features [n, ceil(L / 2**levels)...] of a batch (device tensor)."""
    x = wvfs.signal
    if not x.is_cuda:
        raise _lib.LdspError(-103, "qc_features needs device-resident waveforms (no CPU fallback)")
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    ctx = ctx or _lib.default_context(x.device.index)
    ctx.bind_stream()
    n, L = x.shape
    a, b = -1, -1
    if config is not None:   # the (wvfs, f, config) methods: signalstats(bl_window).mean, shift_waveform  (:26-34, :62-70)
        a, b = window_index(config.bl_window.left, wvfs.t_first, wvfs.dt), window_index(config.bl_window.right, wvfs.t_first, wvfs.dt)
        if not (0 <= a <= b <= L - 1):
            raise WindowError(f"bl_window [{a},{b}] outside a trace of {L} samples")
    lout = _lib.lib().ldsp_qc_features_len(L, int(levels))
    feat = torch.empty((n, lout), dtype=torch.float32, device=x.device)
    norm = torch.empty(n, dtype=torch.float32, device=x.device) if return_norm else None
    _lib.check(_lib.lib().ldsp_qc_features(ctx.handle, C.c_void_p(x.data_ptr()), n, L, int(levels), a, b, C.c_void_p(feat.data_ptr()),
                                           C.c_void_p(norm.data_ptr()) if norm is not None else None))
    return (feat, norm) if return_norm else feat


def get_qc_classifier(wvfs: ArrayOfRDWaveforms, f_evaluate_qc, config: DSPConfig = None, ctx=None):
    """`get_qc_classifier(wvfs, f_evaluate_qc[, config])` — reference src/dsp_ml_routines.jl:9-34 (Haar x 5)."""
    y_pred, _ = f_evaluate_qc(qc_features(wvfs, 5, config, ctx))
    return y_pred


def get_qc_classifier_compressed(wvfs: ArrayOfRDWaveforms, f_evaluate_qc, config: DSPConfig = None, ctx=None):
    """`get_qc_classifier_compressed(wvfs, f_evaluate_qc[, config])` — reference src/dsp_ml_routines.jl:45-70 (Haar x 2)."""
    y_pred, _ = f_evaluate_qc(qc_features(wvfs, 2, config, ctx))
    return y_pred


class RbfSvmPredictor:
    """C-SVC prediction with a radial-basis kernel, the model LIBSVM trains in `get_qc_ml_func` (reference src/ml.jl:6-22).

    One-vs-one voting over the classes exactly as libsvm's `svm_predict_values`: for each class pair (i, j), i < j,
    `dec = sum_{sv of i} coef[j-1][sv] K(sv, x) + sum_{sv of j} coef[i][sv] K(sv, x) - rho[pair]`, a vote for i when
    dec > 0 else for j; the label with most votes (first on ties) wins.  K(u, v) = exp(-gamma |u - v|^2) is evaluated for
    the whole batch as one dense matrix product (rocBLAS through torch) — the only dense contraction on the path.

    support_vectors [nsv, d] grouped by class, n_sv [k] per-class counts, dual_coef [k-1, nsv], rho [k(k-1)/2],
    labels [k] (the layout of libsvm's `svm_model`: SV, nSV, sv_coef, rho, label)."""

    def __init__(self, support_vectors, n_sv, dual_coef, rho, labels, gamma, device="cuda"):
        f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        self.sv, self.coef, self.rho = f(support_vectors), f(dual_coef).reshape(len(labels) - 1, -1), f(rho).reshape(-1)
        self.n_sv = [int(v) for v in n_sv]
        self.labels = torch.as_tensor(labels, dtype=torch.int64, device=device)
        self.gamma = float(gamma)
        self.sv_sq = (self.sv * self.sv).sum(1)

    def __call__(self, features: torch.Tensor):
        x = features.to(self.sv.device, torch.float32)
        d2 = (x * x).sum(1)[:, None] + self.sv_sq[None, :] - 2.0 * (x @ self.sv.T)
        K = torch.exp(-self.gamma * d2.clamp_min(0.0))
        k = len(self.n_sv)
        start = [0]
        for c in self.n_sv:
            start.append(start[-1] + c)
        votes = torch.zeros((x.shape[0], k), dtype=torch.int32, device=x.device)
        dec = []
        p = 0
        for i in range(k):
            for j in range(i + 1, k):
                si, sj = slice(start[i], start[i + 1]), slice(start[j], start[j + 1])
                d = K[:, si] @ self.coef[j - 1, si] + K[:, sj] @ self.coef[i, sj] - self.rho[p]
                votes[:, i] += (d > 0).to(torch.int32)
                votes[:, j] += (d <= 0).to(torch.int32)
                dec.append(d)
                p += 1
        y = self.labels[votes.argmax(dim=1)]   # first maximum on ties, as libsvm
        return y, torch.stack(dec, 1)
