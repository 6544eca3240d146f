"""julia/LegendDSPHIP.jl cannot be executed here (no Julia toolchain), but it can be checked mechanically: its struct
definitions must have the C layout of include/ldsp.h (field order, sizes, offsets — against gcc's offsetof and, when the
library is built, ldsp_abi_sizeof), and every entry point the header declares must be bound by a `ccall` with the
prototype's number of arguments."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "LegendDSPHIP.jl")
HDR = os.path.join(ROOT, "include", "ldsp.h")

PRIM = {"Int32": (4, 4), "Int64": (8, 8), "Float64": (8, 8), "Float32": (4, 4), "Cint": (4, 4)}

# C struct -> (Julia struct, leaf fields in order as C spells them)
PTR48 = ["blmean", "blsigma", "blslope", "bloffset", "tailmean", "tailsigma", "tailslope", "tailoffset", "t0", "t10", "t50", "t80",
         "t90", "t99", "t50_current", "drift_time", "tail_tau", "tail_mean", "tail_sigma", "e_max", "e_min", "e_10410", "e_535",
         "e_313", "e_10410_inv", "e_313_inv", "t0_inv", "e_trap", "e_cusp", "e_zac", "e_trap_max", "e_cusp_max", "e_zac_max",
         "t_trap_max", "t_cusp_max", "t_zac_max", "qdrift", "lq", "a_sg", "a_60", "a_100", "a_raw", "inTrace_intersect",
         "inTrace_n", "n_sat_low", "n_sat_high", "n_sat_low_cons", "n_sat_high_cons"]
SIPM20 = ["t_max", "t_min", "t_max_lar", "t_min_lar", "e_max", "e_min", "e_max_lar", "e_min_lar", "blmean", "blsigma", "blslope",
          "bloffset", "wfmean", "wfsigma", "wfslope", "wfoffset", "threshold", "threshold_DC", "threshold_trap", "threshold_DC_trap"]
TRAP = ["navg", "ngap", "navg2"]
CZ = ["sigma", "flat", "length", "tau", "beta"]
DNI = ["npts", "degree"]
TRIG = ["count", "x", "x_high", "x_tot", "max", "cap", "_pad"]


def sub(prefix, fields):
    return [f"{prefix}.{f}" for f in fields]


C_STRUCTS = {
    "ldsp_trap": ("LdspTrap", TRAP),
    "ldsp_cuspzac": ("LdspCuspZac", CZ),
    "ldsp_dni": ("LdspDni", DNI),
    "ldsp_icpc_params": ("LdspIcpcParams",
                         ["L", "_pad0", "t_first", "dt", "unit_per_us", "sat_low", "sat_high", "bl_from", "bl_until", "tail_from", "tail_until",
                          "pz_c"] + sub("t0_trap", TRAP) + ["t0_mintot", "t0_threshold"] + sub("t0inv_trap", TRAP) + ["tx_mintot"] +
                         sub("int_est", DNI) + ["qdrift_d1", "qdrift_d2", "lq_d1", "lq_d2"] +
                         [f for i in range(3) for f in sub(f"trap_fixed[{i}]", TRAP)] + sub("trap_opt", TRAP) + ["trap_pickoff"] +
                         sub("sig_est", DNI) + sub("cusp", CZ) + sub("zac", CZ) + ["cusp_pickoff", "zac_pickoff", "sg_npts[0]", "sg_npts[1]",
                                                                              "sg_npts[2]", "sg_degree", "cur_left", "cur_right",
                                                                              "intrace_nsigma", "intrace_mintot", "_pad1", "bl_left", "bl_right"]),
    "ldsp_icpc_out": ("LdspIcpcOut", PTR48 + ["stride"]),
    "ldsp_icpc_opts": ("LdspIcpcOpts", ["ext_baseline", "ext_baseline_scale", "main_only", "in_u16"]),
    "ldsp_sipm_params": ("LdspSipmParams",
                         ["L", "_pad0", "t_first", "dt", "unit_per_us", "trunc_from", "trunc_until", "sg_npts", "sg_degree", "sg_mintot", "sg_maxtot",
                          "sg_min_thr", "sg_max_thr", "sg_nsigma", "sg_min_dc_thr", "sg_max_dc_thr", "sg_nsigma_dc", "pz_c"] + sub("trap", TRAP) +
                         ["trap_mintot", "trap_maxtot", "_pad1", "trap_min_thr", "trap_max_thr", "trap_nsigma", "trap_min_dc_thr", "trap_max_dc_thr",
                          "trap_nsigma_dc"]),
    "ldsp_trig_out": ("LdspTrigOut", TRIG),
    "ldsp_sipm_out": ("LdspSipmOut", SIPM20 + sub("trig", TRIG) + sub("trig_DC", TRIG) + sub("trig_trap", TRIG) + sub("trig_DC_trap", TRIG)),
    "ldsp_trapgrid_params": ("LdspTrapGridParams", ["L", "_pad0", "t_first", "dt", "bl_from", "bl_until", "pz_c"] + sub("sig_est", DNI) +
                             ["pick_mode", "tx_mintot", "pick_time"]),
}


def parse_julia_structs():
    src = open(JL).read()
    structs = {}
    for m in re.finditer(r"^struct (\w+)\n(.*?)^end", src, re.S | re.M):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("#")[0].strip()
            if not line:
                continue
            mm = re.match(r"(\w+)::(.+)$", line)
            assert mm, f"unparsed field line in struct {m.group(1)}: {line!r}"
            fields.append((mm.group(1), mm.group(2).strip()))
        structs[m.group(1)] = fields
    return structs


def layout(tname, structs):
    """-> (size, align, [(path, offset, size)]) of a Julia isbits type under C layout rules"""
    if tname in PRIM:
        s, a = PRIM[tname]
        return s, a, [("", 0, s)]
    if tname.startswith("Ptr{"):
        return 8, 8, [("", 0, 8)]
    m = re.match(r"NTuple\{(\d+),\s*(.+)\}$", tname)
    if m:
        n, inner = int(m.group(1)), m.group(2)
        s, a, leaves = layout(inner, structs)
        out = []
        for i in range(n):
            out += [(f"[{i}]{p}", i * s + o, z) for p, o, z in leaves]
        return n * s, a, out
    fields = structs[tname]
    off, align, out = 0, 1, []
    for name, t in fields:
        s, a, leaves = layout(t, structs)
        off = (off + a - 1) // a * a
        out += [(f".{name}{p}", off + o, z) for p, o, z in leaves]
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, out


@pytest.fixture(scope="module")
def c_layout():
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HDR}"', "int main(void) {"]
    for cs, (_, fields) in C_STRUCTS.items():
        lines.append(f'  printf("S {cs} %zu\\n", sizeof({cs}));')
        for f in fields:
            lines.append(f'  printf("F {cs} {f} %zu %zu\\n", offsetof({cs}, {f}), sizeof((({cs}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "off.c"), os.path.join(d, "off")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-std=c11", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    sizes, fields = {}, {}
    for line in out.splitlines():
        parts = line.split()
        if parts[0] == "S":
            sizes[parts[1]] = int(parts[2])
        else:
            fields.setdefault(parts[1], []).append((parts[2], int(parts[3]), int(parts[4])))
    return sizes, fields


def test_julia_structs_have_the_c_layout(c_layout):
    sizes, cfields = c_layout
    structs = parse_julia_structs()
    for cs, (js, _) in C_STRUCTS.items():
        assert js in structs, f"{js} missing from LegendDSPHIP.jl"
        size, _, leaves = layout(js, structs)
        assert size == sizes[cs], (js, size, sizes[cs])
        assert len(leaves) == len(cfields[cs]), (js, len(leaves), len(cfields[cs]))
        for (jpath, joff, jsize), (cname, coff, csize) in zip(leaves, cfields[cs]):
            assert (joff, jsize) == (coff, csize), (js, jpath, joff, jsize, cname, coff, csize)
            # names agree wherever the Julia side spells the field (tuples of pointers stand for the named columns)
            jname = re.sub(r"\[(\d+)\]", r"[\1]", jpath.lstrip("."))
            if not jname.startswith(("cols[", "scalars[")):
                assert jname.replace(".[", "[") == cname, (js, jname, cname)


def test_julia_structs_match_the_built_library():
    so = os.path.join(ROOT, "legenddsp.jl_amd", "csrc", "libldsp_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    import ctypes
    lib = ctypes.CDLL(so)
    lib.ldsp_abi_sizeof.restype = ctypes.c_int64
    structs = parse_julia_structs()
    for which, js in enumerate(("LdspIcpcParams", "LdspIcpcOut", "LdspSipmParams", "LdspSipmOut", "LdspTrigOut", "LdspIcpcOpts")):
        assert layout(js, structs)[0] == lib.ldsp_abi_sizeof(which), js
    src = open(JL).read()
    assert f"const LDSP_ABI_VERSION = {lib.ldsp_abi_version()}" in src


def _c_prototypes():
    hdr = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int32_t|int64_t|const char\*)\s+(ldsp_\w+)\s*\(([^;{]*?)\)\s*;", hdr, re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return protos


def test_every_entry_point_has_a_ccall_with_the_right_arity():
    src = open(JL).read()
    protos = _c_prototypes()
    assert len(protos) >= 44
    for name, nargs in protos.items():
        m = re.search(r"ccall\(\(:%s, libldsp\),\s*[\w{}]+,\s*\((.*?)\)\s*,?\s*(?:\n|\w|\))" % name, src, re.S)
        assert m, f"{name}: no ccall in LegendDSPHIP.jl"
        types = m.group(1).strip()
        # split the argument-type tuple at top-level commas
        depth, cnt, tok = 0, 0, ""
        for ch in types:
            if ch in "{(":
                depth += 1
            elif ch in "})":
                depth -= 1
            if ch == "," and depth == 0:
                cnt += 1 if tok.strip() else 0
                tok = ""
            else:
                tok += ch
        cnt += 1 if tok.strip() else 0
        assert cnt == nargs, f"{name}: header has {nargs} arguments, the ccall passes {cnt} ({types})"
