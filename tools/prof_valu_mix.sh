#!/bin/bash
# tools/prof_valu_mix.sh OUT [LIB] [opts..] — dynamic VALU instruction mix of the dsp_icpc kernel by class (rocprofv3 PMC passes over
# tools/prof_small.py): f32 add / mul / fma / transcendental, f64, int32, int64, conversions; the remainder of SQ_INSTS_VALU is
# moves, selects, compares and DPP copies.  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
if [ -n "$1" ] && [ -f "$1" ]; then export LDSP_HIP_LIB=$(readlink -f $1); shift; fi
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/tools/prof_small.py 65536 "$@" > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "icpc" not in k: continue
        k = k.split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in acc:
    print(k)
    w = acc[k].get("SQ_WAVES", 0) / max(1, len(nd[(k, "SQ_WAVES")]))
    for c in sorted(acc[k]):
        v = acc[k][c] / max(1, len(nd[(k, c)]))
        print("   %-28s %16.0f per dispatch %10.1f per wave" % (c, v, v / w if w else float("nan")))
PY
