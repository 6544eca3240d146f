"""Fixed-seed cases of the randomised parity sweeps (tools/fuzz_icpc.py / tools/fuzz_sipm.py run the same generator
open-ended): dsp_icpc and dsp_sipm through the C ABI against the oracle with randomised filter parameters, tau, window
placement, trace lengths that do and do not fill a tile, noise levels, pile-up, flat-topped and rail-saturated traces,
discharges and ADC quantisation.  tests/fuzz_cases.py holds the generator (seed 1)."""
import numpy as np
import pytest
import torch

import legenddsp_jl_amd as ldsp
import fuzz_cases
import parity

pytestmark = pytest.mark.gpu
SEED = 1


# L = 6000 / 7300 (traces shorter than the 8192 tile with 16-byte rows: the lean kernel's bounded instantiation, shared or separate
# CUSP / ZAC geometry), 8192 (full tile: the lean kernel) and 16384 (the generic kernel); (2, 1) and (2, 5): noise-free traces with a 13-tap SG window, where
# the in-trace pile-up threshold sits on the rounding residue of the baseline (tests/parity.py, inTrace columns)
@pytest.mark.parametrize("seed,it", [(1, 0), (1, 2), (1, 4), (1, 5), (1, 8), (1, 9), (1, 10), (1, 13), (2, 1), (2, 5)])
def test_icpc_randomised_configuration(orc, seed, it):
    n = 256
    L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(seed, it)
    p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    wf = fuzz_cases.icpc_traces(n, L, it, noise)
    tab = ldsp.icpc_run(wf, p)
    torch.cuda.synchronize()
    gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16, strict=False)
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    bad = [l for l in lines if f"bad=0/{n}" not in l]
    assert worst <= parity.FLIP_FRAC, descr + "\n" + "\n".join(bad)


@pytest.mark.parametrize("it", range(8))
def test_icpc_randomised_configuration_odd_lengths_and_long_windows(orc, it):
    """The same generator with trace lengths that are no multiple of four samples (4-byte aligned rows, the last quad of a trace read
    sample by sample) and optimised Savitzky-Golay windows of up to 27 taps (25: the streaming form of the lean kernel) — what the
    lean kernel admits since round 4 — with shared and separate CUSP / ZAC (the second pass rebuilds y)."""
    n = 256
    L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(3, it, wide=True)
    p = ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt)
    wf = fuzz_cases.icpc_traces(n, L, it, noise)
    tab = ldsp.icpc_run(wf, p)
    torch.cuda.synchronize()
    name = ldsp.default_context().last_kernel_name()
    assert name == ("lean3::icpc_lean3_kernel" if p.sg_npts[0] <= 25 else "icpc_kernel"), (descr, name)
    gpu = {k: v.cpu().numpy() for k, v in ldsp.table_columns(tab).items()}
    host = wf.cpu().numpy()
    ora = orc.dsp_icpc(host, p, nthreads=16, strict=False)
    lines, worst = parity.compare(gpu, ora, wf=host, params=p, orc=orc)
    bad = [l for l in lines if f"bad=0/{n}" not in l]
    assert worst <= parity.FLIP_FRAC, descr + " sg_taps=" + str(p.sg_npts[0]) + "\n" + "\n".join(bad)


@pytest.mark.parametrize("it", range(6))
def test_sipm_randomised_configuration(orc, it):
    n = 192
    L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(SEED, it)
    p = ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0)
    wf = fuzz_cases.sipm_traces(n, L, it, noise, mean_pulses)
    sc, trig = ldsp.sipm_run(wf, p)
    torch.cuda.synchronize()
    ora = orc.dsp_sipm(wf.cpu().numpy(), p, nthreads=16)
    msgs = fuzz_cases.sipm_compare(sc, trig, ora, n, wf, p, orc)
    assert not msgs, descr + ": " + "; ".join(msgs)
