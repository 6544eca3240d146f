"""The checker checks: rules of tests/parity.py and tests/sipm_budget.py that ACCEPT a difference are exercised on the CPU against the oracle alone —
what they must accept and what they must keep rejecting.  (The GPU tests rely on them; no kernel runs here.)"""
import numpy as np
import torch

import legenddsp_jl_amd as ldsp
import fuzz_cases
import parity


def _batch(seed, it, wide):
    L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(seed, it, wide=wide)
    p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    wf = ldsp.synth.hpge_batch(256, L, device="cpu", seed=1000 + it, noise=noise)     # (fuzz_cases.icpc_traces, on the host: the generator is counter-based)
    wf[:8] = wf[:8] + torch.roll(wf[8:16] - wf[8:16, :1], 900, dims=1) * 0.5
    wf[16:20] = wf[16:20].clamp(max=65520.0 * 0.1 + 900)
    wf[20:22] = (wf[20:22] * 8).clamp(min=0.0, max=65520.0)
    return wf.numpy(), p


def test_t0_rule_accepts_a_run_broken_at_the_resolution_and_nothing_else(orc):
    """Round 4, sweep of seed 7, case 10, row 59 (profiles/r04_fuzz_summary.txt): behind the crossing of the inverted t0 trapezoid one sample of the
    94-sample run holds by 4e-4; the kernels lose it and report the next run, 51.840 us against the oracle's 51.791 us.  The rule accepts that row —
    the difference is reproduced by deciding that sample the other way — and keeps rejecting shifted crossings: by 0.05 us and 0.02 us (three and one
    samples) and by 0.002 us on the steep t0 crossings of a pulse."""
    host, p = _batch(7, 10, True)
    ora = orc.dsp_icpc(host, p, nthreads=8, strict=False)
    gpu = {k: np.array(v, dtype=np.float64).copy() for k, v in ora.items()}
    assert abs(ora["t0_inv"][59] - 51.79137888868318) < 1e-9
    gpu["t0_inv"][59] = 51.84001159667969                      # what both kernels returned
    for r in range(60, 80):
        gpu["t0_inv"][r] += 0.05
    for r in range(100, 120):
        gpu["t0"][r] += 0.02
    for r in range(130, 140):
        gpu["t0"][r] += 0.002
    bad, _ = parity.bad_mask("t0_inv", gpu, ora, host, p, orc)
    assert not bad[59] and bad[60:80].all() and bad.sum() == 20
    bad, _ = parity.bad_mask("t0", gpu, ora, host, p, orc)
    assert bad[100:120].all() and bad[130:140].all() and bad.sum() == 30
    # without the trace the rule cannot apply: the row stays flagged
    bad, _ = parity.bad_mask("t0_inv", gpu, ora)
    assert bad[59]
