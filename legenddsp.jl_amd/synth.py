"""Synthetic waveform batches — there are no datasets in the
image.  Shapes follow the reference's own fixtures: `make_fake_waveform`
(reference test/test_dsp_icpc.jl:11-32) and `make_sipm_waveform`
(reference test/test_dsp_sipm.jl:10-26), with per-trace randomised baseline,
amplitude, onset, rise time and Gaussian noise.  Generated with torch so the
same code fills host memory (CPU tests) or HBM (bench) directly.

Counter-based: every random number is a hash of (seed, stream, global trace index, sample index) — SplitMix64's
finaliser over a 64-bit counter, in torch int64 arithmetic (which wraps) — so trace i of a batch is the same whatever
the batch it is generated in: `hpge_batch(n, first_trace=k)` IS rows k .. k+n-1 of the whole job's batch, for any k and any
chunking (a rank of the multi-GPU bench generates exactly its shard; a test can regenerate any row of the 1 M-trace batch).
The integer part is bit-identical on CPU and GPU and the Gaussian transform is rounded to float32 from double precision, so a
batch generated on the host equals the one generated on the device (to a last-place difference in about one sample of 1e8).
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


_M64 = (1 << 64) - 1


def _s64(v):
    """Python int -> the same 64 bits as a signed int64 value."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLD, _MIX1, _MIX2, _STREAM = _s64(0x9E3779B97F4A7C15), _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB), 0xD1342543DE82EF95


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)      # logical shift right of an int64 tensor


def _hash_u24(counter, seed, stream):
    """SplitMix64 finaliser of `counter` (int64 tensor) keyed by (seed, stream) -> 24 random bits per element (int64 tensor)."""
    z = counter * _GOLD + _s64(seed * 0x2545F4914F6CDD1D + (stream + 1) * _STREAM)
    z = (z ^ _lsr(z, 30)) * _MIX1
    z = (z ^ _lsr(z, 27)) * _MIX2
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40)


def _uniform(counter, seed, stream):
    """U(0, 1) float32, never 0 or 1."""
    return (_hash_u24(counter, seed, stream).to(torch.float32) + 0.5) * (1.0 / 16777216.0)


def _normal_into(out, first_counter, seed, stream):
    """Fill `out` [m, L] (float32) with N(0, 1): element (i, j) from the counter first_counter + i * L + j (Box-Muller on two
    uniforms of that counter).  The transform runs in float64 and is rounded to float32 once: host and device libraries differ in
    the last place of a double logarithm / cosine, which the rounding hides but for a sample in ~1e8 (checked on the GPU box,
    tests/test_icpc_gpu.py::test_synthetic_batch_is_the_same_on_host_and_device)."""
    m, L = out.shape
    c = torch.arange(m * L, device=out.device, dtype=torch.int64).view(m, L) + int(first_counter)
    u1 = (_hash_u24(c, seed, stream).to(torch.float64) + 0.5) * (1.0 / 16777216.0)
    u2 = (_hash_u24(c, seed, stream + 1).to(torch.float64) + 0.5) * (1.0 / 16777216.0)
    del c
    torch.log(u1, out=u1)
    u1.mul_(-2.0).sqrt_()
    u2.mul_(2.0 * math.pi)
    torch.cos(u2, out=u2)
    u1.mul_(u2)
    out.copy_(u1)
    return out


def _chunk(chunk, dev, L):
    """Traces per pass: `chunk` bounds the temporaries only (the values do not depend on it).  Default: 2^26 samples per pass on
    the host, 2^29 on a GPU (a few 4 GB int64 temporaries; fewer, larger kernels)."""
    return int(chunk) if chunk else max(1, ((1 << 29) if dev.type == "cuda" else (1 << 26)) // L)


def hpge_batch(n, L=8192, seed=0x1E6E4D, device="cpu", noise=3.0, chunk=None, out=None, first_trace=0):
    """x[i,j] = B + A*s(j-j0; R) + noise*g, float32.  B~U[900,1100], A~U[500,2e4],
    j0~U{2950..3050}*L/8192, R~U{60..190}*L/8192, decay 31250*L/8192 samples.  Row i is global trace first_trace + i
    (counter-based, see the module text); `chunk` only bounds the temporaries."""
    dev = torch.device(device)
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=dev)
    sc = L / 8192.0
    chunk = _chunk(chunk, dev, L)
    j = torch.arange(L, device=dev, dtype=torch.float32)[None, :]
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        t = torch.arange(first_trace + c0, first_trace + c1, device=dev, dtype=torch.int64)[:, None]   # global trace index
        B = 900 + 200 * _uniform(t, seed, 1)
        A = 500 + 19500 * _uniform(t, seed, 2)
        j0 = torch.floor((2950 + 101 * _uniform(t, seed, 3)) * sc)
        R = torch.floor((60 + 131 * _uniform(t, seed, 4)) * sc).clamp_(min=1)
        x = out[c0:c1]
        _normal_into(x, (first_trace + c0) * L, seed, 8)
        x.mul_(noise)
        u = j - j0
        ramp = (u / R).clamp_(0, 1)
        dec = torch.exp((u - R).clamp_(min=0).to(torch.float64).mul_(-1.0 / (31250.0 * sc))).to(torch.float32)   # (double, rounded once: host = device)
        x.add_(B + A * ramp * dec)
    return out


_POISSON3_CDF = None


def _poisson_from_uniform(u, mean, kmax):
    """Inverse-CDF Poisson(mean), clamped to kmax."""
    cdf, p, acc = [], math.exp(-mean), 0.0
    for k in range(kmax):
        acc += p
        cdf.append(acc)
        p *= mean / (k + 1)
    edges = torch.tensor(cdf, device=u.device, dtype=torch.float32)
    return (u > edges[None, :]).sum(dim=1, keepdim=True).to(torch.float32)     # u: [m, 1] -> K: [m, 1]


def sipm_batch(n, L=16384, seed=0x51B3, device="cpu", noise=0.3, mean_pulses=3.0, chunk=None, out=None, first_trace=0):
    """K~Poisson(3) pulses of the reference SiPM shape (10-sample rise 1-exp(-k/3),
    decay 30 samples), amplitudes U[2,10], uniform positions, N(0,0.3) noise.  Counter-based like `hpge_batch`."""
    dev = torch.device(device)
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=dev)
    kmax = 12
    chunk = _chunk(chunk, dev, L)
    j = torch.arange(L, device=dev, dtype=torch.float32)[None, :]
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        t = torch.arange(first_trace + c0, first_trace + c1, device=dev, dtype=torch.int64)[:, None]
        x = out[c0:c1]
        _normal_into(x, (first_trace + c0) * L, seed, 8)
        x.mul_(noise)
        K = _poisson_from_uniform(_uniform(t, seed, 1), mean_pulses, kmax)
        for k in range(kmax):
            pos = torch.floor(64 + (L - 256) * _uniform(t, seed, 16 + 2 * k))
            amp = (2 + 8 * _uniform(t, seed, 17 + 2 * k)) * (K > k)
            if not bool((amp > 0).any()):
                continue
            u = j - pos
            rise = (1 - torch.exp(u.clamp(min=0).to(torch.float64).mul_(-1.0 / 3.0)).to(torch.float32)) * ((u >= 0) & (u < 10))
            fall = torch.exp((u - 10).clamp(min=0).to(torch.float64).mul_(-1.0 / 30.0)).to(torch.float32) * (u >= 10)
            x.add_(amp * (rise + fall))
    return out
