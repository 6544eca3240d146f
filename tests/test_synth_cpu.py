"""The synthetic batches are counter-based: trace i depends on (seed, i) only — not on the batch it is generated in, its
chunking, or the rank that generates it (bench.py: rank r generates rows r*n .. (r+1)*n-1 of the whole job's batch)."""
import torch

import legenddsp_jl_amd as ldsp


def test_hpge_rows_do_not_depend_on_the_batch():
    whole = ldsp.synth.hpge_batch(96, 2048, seed=7)
    for first, m, chunk in ((0, 96, 13), (40, 17, 5), (95, 1, 8192)):
        part = ldsp.synth.hpge_batch(m, 2048, seed=7, first_trace=first, chunk=chunk)
        assert torch.equal(part, whole[first:first + m]), (first, m, chunk)
    assert not torch.equal(ldsp.synth.hpge_batch(4, 2048, seed=8), whole[:4])
    # far into a large job (rank 7 of config 4: first trace 8 750 000): the counters are 64-bit
    far = ldsp.synth.hpge_batch(3, 8192, first_trace=8_750_000)
    assert torch.equal(ldsp.synth.hpge_batch(1, 8192, first_trace=8_750_002), far[2:3])
    assert bool(torch.isfinite(far).all())


def test_sipm_rows_do_not_depend_on_the_batch():
    whole = ldsp.synth.sipm_batch(40, 4096, seed=3)
    part = ldsp.synth.sipm_batch(9, 4096, seed=3, first_trace=21, chunk=4)
    assert torch.equal(part, whole[21:30])


def test_noise_moments():
    x = ldsp.synth.hpge_batch(64, 8192, noise=3.0)[:, :2400]          # baseline region: level + noise
    z = (x - x.mean(dim=1, keepdim=True)) / 3.0
    assert abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 3).mean())) < 0.03 and abs(float((z ** 4).mean()) - 3.0) < 0.1
    assert abs(float((z[:, 1:] * z[:, :-1]).mean())) < 0.01           # neighbouring samples uncorrelated
    assert abs(float((z[1:] * z[:-1]).mean())) < 0.01                 # neighbouring traces uncorrelated
