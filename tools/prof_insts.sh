#!/bin/bash
# tools/prof_insts.sh OUTDIR [opt=val ...] — dynamic instruction mix of the dsp_icpc kernels per wave (rocprofv3 PMC, 8192 traces):
# SQ_INSTS_VALU / SALU / LDS / SMEM / VMEM + SQ_WAVES + SQ_WAVE_CYCLES + SQ_BUSY_CYCLES in separate passes of <= 8 SQ counters.
# Honours LDSP_HIP_LIB.  Run on the GPU box (gpurun).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p1 -o p --output-format csv -- python3 $R/tools/prof_small.py 8192 "$@" > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -d $O/p2 -o p --output-format csv -- python3 $R/tools/prof_small.py 8192 "$@" > $O/p2.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
        for r in csv.DictReader(open(f)):
            if "ldsp::" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0][-60:]
        for k, v in per.items():
            for c, x in v.items(): acc[names[k]][c].append(x)
    for k, v in acc.items():
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        print(k, "waves", w)
        for c, x in sorted(v.items()):
            if c != "SQ_WAVES": print("   %-22s per wave %10.1f" % (c, sum(x) / len(x) / w))
PY
