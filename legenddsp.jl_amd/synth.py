"""Synthetic waveform batches — there are no datasets in the
image.  Shapes follow the reference's own fixtures: `make_fake_waveform`
(reference test/test_dsp_icpc.jl:11-32) and `make_sipm_waveform`
(reference test/test_dsp_sipm.jl:10-26), with per-trace randomised baseline,
amplitude, onset, rise time and Gaussian noise.  Generated with torch so the
same code fills host memory (CPU tests) or HBM (bench) directly.
"""
import math

import torch


def reference_hpge_waveform(n=8192, dtype=torch.float64):
    """The noiseless trace of test/test_dsp_icpc.jl:11-32 (1-based i in the reference)."""
    i = torch.arange(1, n + 1, dtype=torch.float64)
    b_end, r_end = round(48e3 / 16), round(50e3 / 16)
    amp, tau, off = 10000.0, 500e3 / 16, 1000.0
    sig = torch.where(i < b_end, torch.full_like(i, off),
                      torch.where(i < r_end, off + amp * (i - b_end) / (r_end - b_end),
                                  off + amp * torch.exp(-(i - r_end) / tau)))
    return sig.to(dtype)


def reference_sipm_waveform(n=6250, dtype=torch.float64):
    """The noiseless trace of test/test_dsp_sipm.jl:10-26."""
    i = torch.arange(1, n + 1, dtype=torch.float64)
    start, width, amp, tau = round(50e3 / 16), 10, 5.0, 30.0
    rise = amp * (1 - torch.exp(-(i - start) / 3.0))
    fall = amp * torch.exp(-(i - start - width) / tau)
    sig = torch.where((i >= start) & (i < start + width), rise,
                      torch.where(i >= start + width, fall, torch.zeros_like(i)))
    return sig.to(dtype)


def hpge_batch(n, L=8192, seed=0x1E6E4D, device="cpu", noise=3.0, chunk=65536, out=None, first_trace=0):
    """x[i,j] = B + A*s(j-j0; R) + noise*g, float32.  B~U[900,1100], A~U[500,2e4],
    j0~U{2950..3050}*L/8192, R~U{60..190}*L/8192, decay 31250*L/8192 samples."""
    dev = torch.device(device)
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=dev)
    sc = L / 8192.0
    j = torch.arange(L, device=dev, dtype=torch.float32)[None, :]
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 7919 * ((first_trace + c0) // chunk))
        m = c1 - c0
        B = 900 + 200 * torch.rand(m, 1, generator=g, device=dev)
        A = 500 + 19500 * torch.rand(m, 1, generator=g, device=dev)
        j0 = torch.floor((2950 + 101 * torch.rand(m, 1, generator=g, device=dev)) * sc)
        R = torch.floor((60 + 131 * torch.rand(m, 1, generator=g, device=dev)) * sc).clamp_(min=1)
        x = out[c0:c1]
        torch.randn((m, L), generator=g, device=dev, out=x)
        x.mul_(noise)
        u = j - j0
        ramp = (u / R).clamp_(0, 1)
        dec = torch.exp(-(u - R).clamp_(min=0) / (31250.0 * sc))
        x.add_(B + A * ramp * dec)
    return out


def sipm_batch(n, L=16384, seed=0x51B3, device="cpu", noise=0.3, mean_pulses=3.0, chunk=16384, out=None, first_trace=0):
    """K~Poisson(3) pulses of the reference SiPM shape (10-sample rise 1-exp(-k/3),
    decay 30 samples), amplitudes U[2,10], uniform positions, N(0,0.3) noise."""
    dev = torch.device(device)
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=dev)
    kmax = 12
    j = torch.arange(L, device=dev, dtype=torch.float32)[None, :]
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 104729 * ((first_trace + c0) // chunk))
        m = c1 - c0
        x = out[c0:c1]
        torch.randn((m, L), generator=g, device=dev, out=x)
        x.mul_(noise)
        K = torch.poisson(torch.full((m, 1), mean_pulses, device=dev), generator=g).clamp_(max=kmax)
        for k in range(kmax):
            pos = torch.floor(64 + (L - 256) * torch.rand(m, 1, generator=g, device=dev))
            amp = (2 + 8 * torch.rand(m, 1, generator=g, device=dev)) * (K > k)
            u = j - pos
            rise = (1 - torch.exp(-u.clamp(min=0) / 3.0)) * ((u >= 0) & (u < 10))
            fall = torch.exp(-(u - 10).clamp(min=0) / 30.0) * (u >= 10)
            x.add_(amp * (rise + fall))
    return out
