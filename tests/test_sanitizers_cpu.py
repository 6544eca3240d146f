"""AddressSanitizer + UndefinedBehaviorSanitizer leg (CPU only; GPU sanitizers are not available on this pool):
  * the oracle (oracle/ldsp_oracle.c) driven by tests/sanitize/oracle_driver.c over the reference configuration and the
    randomised configurations of tests/fuzz_cases.py, on traces that include flat / clipped / negative / spiky ones, plus every
    extractor and filter at its boundary arguments;
  * the host side of the C ABI (csrc/ldsp_api.hip compiled as plain C++ by g++, launchers stubbed): the complete lowering of
    parameter blocks (ldsp_icpc_check_params) for valid blocks and for blocks with fields corrupted at random (what a
    binding that mis-lays the struct would send), and the coefficient entry points;
  * the bit-mask run scans of the kernels (csrc/run_scan.hpp, integer code that g++ compiles as it is) against a bit-by-bit scan.
A sanitizer report aborts the driver (-fno-sanitize-recover), which fails the test."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import legenddsp_jl_amd as ldsp
import fuzz_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _valid_icpc_blocks():
    blocks = [ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, 8192, 0.0, 16.0)]
    for it in range(14):
        L, dt, cfg, tau, pf, noise, descr = fuzz_cases.icpc_case(1, it)
        if L <= 8192:                       # keeps the oracle leg short
            blocks.append(ldsp.lower_icpc(cfg, tau, pf, L, 0.0, dt))
    return blocks


def _sipm_blocks():
    blocks = [ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, 6250, 0.0, 16.0)]
    for it in range(6):
        L, cfg, pf, noise, mean_pulses, descr = fuzz_cases.sipm_case(1, it)
        if L <= 8192:
            blocks.append(ldsp.lower_sipm(cfg, pf, L, 0.0, 16.0))
    return blocks


def _corrupted(blocks, n, seed):
    """copies of valid blocks with a few 4-byte fields overwritten: small / huge / negative integers, NaN / inf / zero floats"""
    rng = np.random.default_rng(seed)
    ints = np.array([0, 1, -1, 2, 7, 64, 8191, 8192, 40000, 2**31 - 1, -2**31], dtype=np.int64)
    flts = np.array([0.0, -1.0, 1e-30, 1e30, np.inf, -np.inf, np.nan], dtype=np.float32)
    out = []
    for k in range(n):
        b = bytearray(bytes(blocks[k % len(blocks)]))
        for _ in range(int(rng.integers(1, 4))):
            off = 4 * int(rng.integers(0, len(b) // 4))
            if rng.random() < 0.5:
                b[off:off + 4] = int(rng.choice(ints)).to_bytes(4, "little", signed=True)
            else:
                b[off:off + 4] = np.float32(rng.choice(flts)).tobytes()
        out.append(bytes(b))
    return out


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("gcc / g++ not available")
    return tmp_path_factory.mktemp("san")


def test_oracle_under_asan_ubsan(workdir):
    exe = str(workdir / "oracle_driver")
    subprocess.check_call(["gcc", "-std=gnu11", "-fopenmp", "-ffp-contract=off"] + SAN +
                          [os.path.join(ROOT, "tests/sanitize/oracle_driver.c"), os.path.join(ROOT, "oracle/ldsp_oracle.c"), "-lm", "-o", exe])
    icpc, sipm = str(workdir / "icpc.bin"), str(workdir / "sipm.bin")
    with open(icpc, "wb") as f:
        for b in _valid_icpc_blocks():
            f.write(bytes(b))
    with open(sipm, "wb") as f:
        for b in _sipm_blocks():
            f.write(bytes(b))
    r = subprocess.run([exe, icpc, sipm], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("done"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith(("icpc", "sipm"))]
    assert len(lines) >= 8 and all(l.split()[3] == "0" for l in lines if "rc" in l), lines     # every valid block runs clean


def test_host_lowering_under_asan_ubsan(workdir):
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip/hip_runtime.h")):
        pytest.skip("HIP headers not found")
    obj, exe = str(workdir / "ldsp_api_san.o"), str(workdir / "host_driver")
    common = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc] + SAN
    subprocess.check_call(common + ["-x", "c++", "-c", os.path.join(ROOT, "legenddsp.jl_amd/csrc/ldsp_api.hip"), "-o", obj])
    subprocess.check_call(common + [os.path.join(ROOT, "tests/sanitize/host_driver.cpp"), obj, "-L/opt/rocm/lib", "-lamdhip64",
                                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    valid = _valid_icpc_blocks()
    blobs = [bytes(b) for b in valid] + _corrupted(valid, 2000, seed=7)
    path = str(workdir / "blocks.bin")
    with open(path, "wb") as f:
        for b in blobs:
            f.write(b)
    r = subprocess.run([exe, path], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith(f"done {len(blobs)}"), (r.returncode, r.stdout[-1000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    rcs = [int(l.split()[1]) for l in r.stdout.splitlines() if l and l[0].isdigit()]
    assert all(rc == 0 for rc in rcs[:len(valid)])              # what Python lowers, the host lowering accepts
    assert any(rc != 0 for rc in rcs[len(valid):])              # and corrupted blocks are rejected, not executed


def test_run_scans_under_asan_ubsan(workdir):
    """csrc/run_scan.hpp (the Intersect run scans of the kernels, plain integer code) compiled by g++: the word-by-word and the
    loop-free forms against a bit-by-bit scan — tests/sanitize/run_scan_driver.cpp"""
    exe = str(workdir / "run_scan_driver")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "legenddsp.jl_amd/csrc")] + SAN +
                          [os.path.join(ROOT, "tests/sanitize/run_scan_driver.cpp"), "-o", exe])
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "run_scan: 0 mismatches" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
