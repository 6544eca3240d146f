#!/bin/bash
# A/B timing of the current build against /tmp-style reference builds passed as arguments (paths inside the repo), plus
# the icpc / sipm parity tests on the current build.  Usage: bash tools/quick_ab.sh [build/ref.so ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for so in "$@" ""; do
  echo "== ${so:-current}"
  LDSP_HIP_LIB=${so:+$R/$so} python3 $R/tools/gpu_time.py 65536 2>&1 | grep -v amdgpu.ids
  LDSP_HIP_LIB=${so:+$R/$so} python3 $R/tools/gpu_time_sipm.py 32768 2>&1 | grep "dsp_sipm n="
done
python3 -m pytest $R/tests/test_icpc_gpu.py $R/tests/test_sipm_gpu.py $R/tests/test_functors_gpu.py -x -q 2>&1 | tail -3
