"""CPU-side checks of the C-ABI library: it loads, exports every symbol
include/ldsp.h declares, and its struct layouts match the ctypes mirror."""
import ctypes
import os
import re

import legenddsp_jl_amd as ldsp
from legenddsp_jl_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    ldsp.build()
    hdr = open(os.path.join(ROOT, "include", "ldsp.h")).read()
    declared = set(re.findall(r"\b(ldsp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ldsp_ctx"}
    lib = ctypes.CDLL(_lib._SO)
    missing = sorted(s for s in declared if not hasattr(lib, s))
    assert not missing, f"declared in ldsp.h but not exported: {missing}"
    assert declared == set(_lib.DECLARED_SYMBOLS), declared ^ set(_lib.DECLARED_SYMBOLS)


def test_struct_sizes_match():
    lib = _lib.lib()  # raises on ABI / size mismatch
    assert lib.ldsp_abi_version() == _abi.LDSP_ABI_VERSION
    for which, st in enumerate((_abi.IcpcParams, _abi.IcpcOut, _abi.SipmParams, _abi.SipmOut, _abi.TrigOut)):
        assert lib.ldsp_abi_sizeof(which) == ctypes.sizeof(st)


def test_coefficient_builders_match_oracle(orc):
    import numpy as np
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, 16.0)
    lib = _lib.lib()
    for name, oc in (("ldsp_cusp_coeffs", orc.cusp_coeffs), ("ldsp_zac_coeffs", orc.zac_coeffs)):
        cz = p.cusp if "cusp" in name else p.zac
        h = np.empty(cz.length)
        assert getattr(lib, name)(ctypes.byref(cz), h.ctypes.data_as(ctypes.c_void_p)) == 0
        np.testing.assert_allclose(h, oc(cz), rtol=1e-12, atol=1e-15)
    for npts in (5, 7, 13):
        h = np.empty(npts)
        assert lib.ldsp_sg_coeffs(npts, 3, 1, h.ctypes.data_as(ctypes.c_void_p)) == 0
        np.testing.assert_allclose(h, orc.sg_coeffs(npts, 3, 1), rtol=1e-10, atol=1e-14)
    # quadratic/cubic SG first derivative, 5 points: the textbook (-2,-1,0,1,2)/10 for degree 2
    h = np.empty(5)
    lib.ldsp_sg_coeffs(5, 2, 1, h.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_allclose(h[::-1], np.array([-2, -1, 0, 1, 2]) / 10, atol=1e-14)


def test_no_cpu_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LdspError):
        ldsp.Context()
    p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, 16.0)
    with pytest.raises(_lib.LdspError):
        ldsp.icpc_run(torch.zeros(2, 8192), p)
