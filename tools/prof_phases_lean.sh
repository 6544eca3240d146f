#!/bin/bash
# tools/prof_phases_lean.sh OUT [LIB] — per-phase dynamic instruction counts of icpc_lean3_kernel (diagnostic build with -DLDSP_DSTOP,
# e.g. tools/dev_build.sh ds -DLDSP_DSTOP): one rocprofv3 PMC pass per stop, cumulative VALU / SALU / LDS / SMEM / BRANCH per wave.
# Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
export LDSP_HIP_LIB=$(readlink -f ${2:-$R/build/dev/libldsp_ds.so})
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STOPS="1 2 3 4 5 6 7 9 11 13 14 15 16 17 18 19 20 21 22 0"
for k in $STOPS; do
  stop=$((100 + k)); [ $k = 0 ] && stop=0
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS -d $O/s$k -o p --output-format csv -- python3 $R/tools/prof_small.py 4096 dbg_stop=$stop > $O/s$k.log 2>&1
done
python3 - <<PY
import csv, collections, glob
prev = None
print("%-6s %9s %9s %9s %9s %9s %9s   (cumulative per wave; increments in parentheses; ALL = SQ_INSTS: every instruction issued, s_nop / s_waitcnt / s_barrier included)" % ("stop", "VALU", "SALU", "LDS", "SMEM", "BRANCH", "ALL"))
for k in [int(x) for x in "$STOPS".split()]:
    acc = collections.defaultdict(float)
    for f in glob.glob("$O/s%d/**/*counter_collection.csv" % k, recursive=True):
        for r in csv.DictReader(open(f)):
            if "icpc_lean3_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    w = acc["SQ_WAVES"] or 1
    cur = [acc[c] / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS")]
    inc = [a - b for a, b in zip(cur, prev)] if prev else cur
    print("%-6s " % ("end" if k == 0 else k) + " ".join("%7.0f(%+5.0f)" % (a, b) for a, b in zip(cur, inc)))
    prev = cur
PY
