#!/bin/bash
# tools/prof_phases_sipm.sh OUT — per-phase dynamic instruction counts and LDS stall counters of k_sipm_s4 (dbg_stop = 1..7, 0):
# one rocprofv3 PMC pass per stop and counter set, cumulative per wave.  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 0; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/a$k -o p --output-format csv -- python3 $R/tools/prof_small_sipm.py 4096 $k > $O/a$k.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/b$k -o p --output-format csv -- python3 $R/tools/prof_small_sipm.py 4096 $k > $O/b$k.log 2>&1 || echo "pass b$k failed"
done
python3 - <<PY
import csv, collections, glob
names = {1: "load+extremes", 2: "SG", 3: "MAD(SG)", 4: "mask+trig(SG)", 5: "integrate+stats", 6: "2x(MAD+trig) DC", 7: "InvCR+trap", 0: "MAD+trig(trap)"}
for tag, cols in (("a", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD")),
                  ("b", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT"))):
    prev = None
    print(("%-22s" + " %16s" * len(cols)) % (("after",) + tuple(c.replace("SQ_", "") for c in cols)))
    for k in [1, 2, 3, 4, 5, 6, 7, 0]:
        acc = collections.defaultdict(float)
        for f in glob.glob("$O/%s%d/**/*counter_collection.csv" % (tag, k), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_sipm" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"])
        w = acc["SQ_WAVES"] or 1
        cur = [acc[c] / w for c in cols]
        inc = [a - b for a, b in zip(cur, prev)] if prev else cur
        print("%-22s " % names[k] + " ".join("%8.0f(%+7.0f)" % (a, b) for a, b in zip(cur, inc)))
        prev = cur
PY
