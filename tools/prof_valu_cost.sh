#!/bin/bash
# tools/prof_valu_cost.sh OUT — price list of gfx950 vector instructions at 6 waves per SIMD (tools/micro/valu_cost.hip): the timing table
# of a plain run, then SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU per kind from one rocprofv3 PMC pass.  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
BIN=$R/tools/micro/valu_cost
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/micro/valu_cost.hip -o $BIN
$BIN > $O/timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM -d $O/pmc -o p --output-format csv -- $BIN pmc > $O/pmc.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv, collections, glob, re
acc = collections.defaultdict(dict)
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"vc<(\d+)>", r["Kernel_Name"])
        if m: acc[int(m.group(1))][r["Counter_Name"]] = float(r["Counter_Value"])
names = {}
for line in open("$O/timing.txt"):
    m = re.match(r"K(\d+)\s+(.*?)\s+n=\s*(\d+)\s+ticks/instr/SIMD =\s*([\d.]+)", line)
    if m: names[int(m.group(1))] = (m.group(2), int(m.group(3)), float(m.group(4)))
with open("$O/table.txt", "w") as out:
    out.write("# kind | instructions per block | wall ticks per instruction and SIMD (6 waves per SIMD) | SQ_ACTIVE_INST_VALU per instruction (quad-cycles) | x4 = cycles\n")
    for k in sorted(names):
        nm, n, t = names[k]
        c = acc.get(k, {})
        iv = c.get("SQ_INSTS_VALU", 0.0); av = c.get("SQ_ACTIVE_INST_VALU", 0.0)
        # subtract the fixed part (prologue: 37 instructions per wave) through the instruction count: ratio of totals is close enough at 200 x n
        ratio = av / iv if iv else float("nan")
        out.write("K%-3d %-52s n=%2d  wall %6.2f   active/inst %6.3f   cycles %6.2f\n" % (k, nm, n, t, ratio, 4 * ratio))
print(open("$O/table.txt").read())
PY
