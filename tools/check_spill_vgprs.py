"""tools/check_spill_vgprs.py [file.hip ...] — compiles the kernels' translation units to gfx950 assembly and reports, per kernel, registers, SGPR / VGPR
spill counts and scratch bytes, and FAILS when a VGPR that holds spilled SGPRs ("SGPR spill to VGPR lane" in the assembly) is itself stored to scratch.
Round 4: a k_sipm_s4 instantiation with ~250 SGPR spills AND ~80 VGPR spills computed a wrong wfslope (every other column right) in exactly one
tile shape; with the bounds' masks no longer shared between stages (stage-local thread index) the spills — and the wrong column — went away.
hipcc saves such a lane-VGPR under the EXEC mask of the moment; lanes that are inactive there lose their SGPRs.  Not part of the test suite
(minutes of compile time); run it after changes that raise register pressure."""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(R, "legenddsp.jl_amd", "csrc")
files = sys.argv[1:] or [os.path.join(CS, f) for f in ("icpc_lean3.hip", "icpc_lean.hip", "icpc_kernel.hip", "sipm_kernel.hip", "functor_kernels.hip")]
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None".split()
bad = 0
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", f, "-o", t.name], check=True, stderr=subprocess.DEVNULL)
        s = open(t.name).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
        g = lambda k: (re.search(k + r":\s*(\d+)", m.group(2)) or [0, "?"])[1]
        meta[m.group(1)] = (g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta:
            continue
        # (the registers hipcc reserves for spilled SGPRs are announced in the assembly; v_writelane_b32 alone is also what the kernels' own
        # ballot collection uses)
        lane = {"v" + n for n in re.findall(r"implicit-def: \$vgpr(\d+) : SGPR spill to VGPR lane", body)}
        stored = set()
        for st in re.findall(r"scratch_store_dword\w* off, (v\d+|v\[\d+:\d+\])", body):
            if st.startswith("v["):
                a, b = map(int, st[2:-1].split(":")); stored |= {f"v{i}" for i in range(a, b + 1)}
            else:
                stored.add(st)
        hit = sorted(lane & stored)
        v, ss, vs, sc = meta[name]
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][-70:]
        if hit or int(vs) or int(ss) > 32:
            print(f"{os.path.basename(f):20s} {short:70s} vgpr {v:>3} sgpr-spills {ss:>3} vgpr-spills {vs:>3} scratch {sc:>4} B" + (f"   LANE-VGPR SPILLED: {hit}" if hit else ""))
        bad += bool(hit)
print("kernels with a spilled SGPR-spill VGPR:", bad)
sys.exit(1 if bad else 0)
