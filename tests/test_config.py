"""Host logic: DSPConfig surface and its lowering (banker's rounding, windows)."""
import pytest

import legenddsp_jl_amd as ldsp
from legenddsp_jl_amd import config as C
from legenddsp_jl_amd.config import us, ns


def test_round_half_even_matches_julia_round():
    # 39us/16ns = 2437.5 -> 2438 ; 5us/16ns = 312.5 -> 312 ; 3us/16ns = 187.5 -> 188
    assert C.nsamples(39 * us, 16) == 2438
    assert C.nsamples(5 * us, 16) == 312
    assert C.nsamples(3 * us, 16) == 188
    assert C.nsamples(40 * ns, 16) == 2 and C.nsamples(100 * ns, 16) == 6
    assert C.round_half_even(0.5) == 0 and C.round_half_even(1.5) == 2 and C.round_half_even(-0.5) == 0


def test_lowering_matches_survey_numbers():
    # SURVEY §8 preamble, 1-based there / 0-based here
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * us, {}, 8192, 0.0, 16.0)
    assert (p.bl_from, p.bl_until) == (0, 2438)
    assert (p.tail_from, p.tail_until) == (4375, 6875)
    assert p.pz_c == pytest.approx(1 / 31250)
    assert repr(p.t0_trap) == "Trap(2,6,125)" and p.t0_mintot == 94 and p.tx_mintot == 2 and p.intrace_mintot == 6
    assert [repr(p.trap_fixed[i]) for i in range(3)] == ["Trap(625,250,625)", "Trap(312,188,312)", "Trap(188,62,188)"]
    assert repr(p.trap_opt) == "Trap(312,156,312)"
    assert (p.cusp.length, p.cusp.flat, p.cusp.sigma, p.cusp.beta) == (2375, 156, 312.5, 2375.0)
    assert (p.sig_est.npts, p.int_est.npts) == (44, 6)
    assert p.sat_high == 65520.0
    assert (p.qdrift_d1, p.qdrift_d2) == (2500.0, 5000.0)


def test_get_fltpars_fallback():
    cfg = ldsp.reference_test_icpc_config()
    assert ldsp.get_fltpars({}, "trap", cfg) == (5 * us, 2.5 * us)
    assert ldsp.get_fltpars({"trap": {"rt": 8 * us}}, "trap", cfg) == (8 * us, 2.5 * us)
    assert ldsp.get_fltpars({}, "sg", cfg) == 100 * ns
    assert ldsp.get_fltpars({"sg": {"wl": 180 * ns}}, "sg", cfg) == 180 * ns
    p = ldsp.lower_icpc(cfg, 500 * us, {"trap": {"rt": 8 * us, "ft": 3 * us}, "sg": {"wl": 180 * ns}}, 8192, 0.0, 16.0)
    assert repr(p.trap_opt) == "Trap(500,188,500)" and p.sg_npts[0] == 11 and p.sg_npts[2] == 7


def test_window_assert_like_reference():
    # config 1 (L = 4096) cannot use the default windows at 16 ns (tailstats.jl:24 @assert): needs 32 ns
    cfg = ldsp.reference_test_icpc_config()
    with pytest.raises(ldsp.WindowError):
        ldsp.lower_icpc(cfg, 500 * us, {}, 4096, 0.0, 16.0)
    p = ldsp.lower_icpc(ldsp.plumbing_icpc_config_4096(), 500 * us, {}, 4096, 0.0, 32.0)
    assert p.tail_until == 3438 and list(p.sg_npts) == [3, 3, 3]
    with pytest.raises(ValueError):      # cubic through 3 points: unsupported, as in the reference
        ldsp.lower_icpc(cfg, 500 * us, {}, 4096, 0.0, 32.0)


def test_sipm_lowering():
    p = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ns}}, 6250, 0.0, 16.0)
    assert p.sg_npts == 13 and (p.sg_mintot, p.sg_maxtot) == (4, 9)
    assert repr(p.trap) == "Trap(6,3,6)" and (p.trap_mintot, p.trap_maxtot) == (3, 16)
    assert (p.trunc_from, p.trunc_until) == (2938, 3312)
    assert p.pz_c == pytest.approx(16 / 3000)


def test_time_axis_filter_reference_cases():
    """reference test/test_timeaxis.jl:7-55: the step is overwritten, the first point moves by the offset, the samples stay; a batch
    gives what a single waveform gives, with one shared axis.  (Metadata only: runs without a device.)"""
    import numpy as np
    import torch
    import legenddsp_jl_amd as ldsp
    rng = np.random.default_rng(7)
    old_offset, old_step, new_offset, new_step = rng.random(4)
    wf = ldsp.ArrayOfRDWaveforms(torch.from_numpy(rng.random((1, 100)).astype(np.float32)), float(old_offset), float(old_step))
    out = ldsp.TimeAxisFilter(float(new_step), float(new_offset))(wf)
    assert out.t_first == float(old_offset) + float(new_offset) and out.dt == float(new_step)          # :24-25
    assert torch.equal(out.signal, wf.signal) and out.nsamples == 100
    to0 = ldsp.TimeAxisFilter(4.0, -wf.t_first)(wf)                                                    # :28-32
    assert to0.t_first == 0.0 and to0.dt == 4.0
    wfs = ldsp.ArrayOfRDWaveforms(torch.from_numpy(rng.random((10, 100)).astype(np.float32)), 0.0, 16.0)   # :35-53
    flt = ldsp.TimeAxisFilter(4.0, 100.0)
    new = flt(wfs)
    one = flt(ldsp.ArrayOfRDWaveforms(wfs.signal[:1], 0.0, 16.0))
    assert len(new) == 10 and (new.t_first, new.dt) == (one.t_first, one.dt) == (100.0, 4.0)
    assert torch.equal(new.signal[:1], one.signal)
    fi = ldsp.fltinstance(flt, ldsp.smplinfo(wfs))                                                     # the protocol functions
    assert ldsp.flt_output_length(fi) == ldsp.flt_input_length(fi) == 100 and ldsp.flt_output_time_axis(fi) == (100.0, 4.0)
    y = torch.empty_like(wfs.signal)
    assert torch.equal(ldsp.rdfilt_(y, fi, wfs.signal), wfs.signal)
