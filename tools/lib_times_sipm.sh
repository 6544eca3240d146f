#!/bin/bash
# tools/lib_times_sipm.sh LIB [LIB ...] — dsp_sipm kernel time of each library build on config 5's shape (tools/gpu_time_sipm.py, first line).
# (A/B of builds made at different times: the staleness check of legenddsp_jl_amd._lib is waived explicitly)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  echo "$lib: $(LDSP_ALLOW_STALE=1 LDSP_HIP_LIB=$(readlink -f $lib) timeout -k 10 200 python3 $R/tools/gpu_time_sipm.py ${LDSP_SIPM_N:-131072} 16384 2>&1 | grep "dsp_sipm n=")"
done
