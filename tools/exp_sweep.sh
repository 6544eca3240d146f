#!/bin/bash
# times the dsp_icpc kernel with every build/exp/libldsp_*.so (tuning experiments), then the default build
R=${GRAFT_REPO_ROOT:-$(pwd)}
for so in $R/build/exp/libldsp_*.so ""; do
  echo "== ${so:-default}"
  LDSP_HIP_LIB=$so python3 $R/tools/gpu_time.py 65536 2>&1 | grep -v amdgpu.ids
done
