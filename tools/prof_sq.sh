#!/bin/bash
# tools/prof_sq.sh OUT [LIB] [opts..] — SQ activity / instruction-cache / memory-path counters of the dsp_icpc kernels (LDSP_PROF_N traces, default 65536; pz_only=1: config 2 only) (several rocprofv3 PMC passes over
# tools/prof_small.py).  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
if [ -n "$1" ] && [ -f "$1" ]; then export LDSP_HIP_LIB=$(readlink -f $1); shift; fi
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum TCC_EA_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum GRBM_GUI_ACTIVE TCC_EA_RD_STALL_sum"; do
  i=$((i+1))
  echo "group $i: $grp"
  rocprofv3 --kernel-trace --kernel-include-regex "icpc|pz_trap" --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/tools/prof_small.py ${LDSP_PROF_N:-65536} "$@" > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "icpc_lean" not in k and "pz_trap" not in k: continue
        k = "lean3::icpc_lean3_kernel" if "lean3" in k else "lean::pz_trap_lean_kernel" if "pz_trap" in k else "lean::icpc_lean_kernel"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-24s %16.0f per dispatch" % (c, acc[k][c] / max(1, len(nd[(k, c)]))))
PY
