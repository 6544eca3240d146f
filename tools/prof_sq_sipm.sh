#!/bin/bash
# tools/prof_sq_sipm.sh OUT [LIB] — SQ activity / wait / LDS counters and the instruction mix of k_sipm_s4 (rocprofv3 PMC passes over
# tools/prof_small_sipm.py, LDSP_PROF_N traces, default 32768) -> gpurun_out/OUT/summary.txt.  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
if [ -n "$1" ] && [ -f "$1" ]; then export LDSP_HIP_LIB=$(readlink -f $1); shift; fi
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "k_sipm" --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/tools/prof_small_sipm.py ${LDSP_PROF_N:-32768} > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY > $O/summary.txt
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in acc:
    print(k)
    w = acc[k].get("SQ_WAVES", 0) / max(1, len(nd[(k, "SQ_WAVES")]))
    for c in sorted(acc[k]):
        v = acc[k][c] / max(1, len(nd[(k, c)]))
        print("   %-28s %16.0f per dispatch %12.1f per wave" % (c, v, v / w if w else float("nan")))
PY
cat $O/summary.txt
