#!/bin/bash
# tools/phase_asm.sh [A B] — assembly of the lean kernel <512,7>; with two phase numbers prints the instructions between the
# "; LDSP_PHASE A" and "; LDSP_PHASE B" markers, otherwise a per-phase static instruction count.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/asm2 && cd /tmp/asm2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function -DLDSP_DEV_512 -save-temps -c $R/legenddsp.jl_amd/csrc/icpc_lean.hip -o lean.o 2>/dev/null
awk '/^_ZN4ldsp4lean16icpc_lean_kernelILi512ELi7E.*:/{f=1} f{print} /^\.Lfunc_end/{if(f) exit}' icpc_lean-hip-amdgcn-amd-amdhsa-gfx950.s > l7.s
if [ -n "$2" ]; then
  awk -v a="LDSP_PHASE $1\$" -v b="LDSP_PHASE $2\$" '$0 ~ a {f=1} f{print} $0 ~ b {if (f) exit}' l7.s | grep -vE "^\s*;|^$|^\s*\."
else
  awk '/LDSP_PHASE/{p=$NF} /^[ \t]+(v_|s_|ds_|global_|buffer_|scratch_)/{n[p]++; t=$1; if (t ~ /^v_/) v[p]++; else if (t ~ /^ds_/) l[p]++; else s[p]++} END{for (k in n) printf "after %3s: %5d  (v %4d  s %4d  ds %4d)\n", k, n[k], v[k], s[k], l[k]}' l7.s | sort -k2 -n
fi
