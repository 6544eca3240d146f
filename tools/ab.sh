#!/bin/bash
# tools/ab.sh OUT LIB [LIB ...] — A/B of library builds on the bench configuration: kernel rate of each (tools/lean_time.py, 262144 traces),
# then parity of the LAST one against the oracle on 512 traces (tools/dev_time.py).  Run on the GPU box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $(dirname $O)
: > $O.txt
for rep in 1 2; do
  for lib in "$@"; do
    LDSP_ALLOW_STALE=1 LDSP_HIP_LIB=$(readlink -f $lib) timeout -k 10 200 python3 $R/tools/lean_time.py 262144 6 2>&1 | tail -1 | tee -a $O.txt
  done
done
last="${@: -1}"
LDSP_ALLOW_STALE=1 LDSP_HIP_LIB=$(readlink -f $last) timeout -k 10 300 python3 $R/tools/dev_time.py 65536 2>&1 | tail -40 | tee -a $O.txt
