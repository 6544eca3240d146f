#!/bin/bash
# tools/l3_asm.sh [extra hipcc flags] — ISA of icpc_lean3_kernel<512, 7, false, true> into /tmp/l3_512.s; prints its register / scratch use
# and a static count of VALU, moves, selects, compares, SALU, LDS.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function -DLDSP_DEV_512 -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
  -S --cuda-device-only $R/legenddsp.jl_amd/csrc/icpc_lean3.hip -o /tmp/l3_all.s 2>/dev/null
awk '/^_ZN4ldsp5lean317icpc_lean3_kernelILi512ELi7ELb0ELb1EEE[A-Za-z0-9_]*:/{f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/l3_all.s > /tmp/l3_512.s
awk '/^_ZN4ldsp5lean317icpc_lean3_kernelILi512ELi7ELb0ELb1EEE[A-Za-z0-9_]*:/{f=1} f&&/\.amdhsa_next_free_vgpr|\.amdhsa_next_free_sgpr|private_segment_fixed_size|; ScratchSize|; Occupancy|sgpr_spill|vgpr_spill/{print} f&&/\.end_amdhsa_kernel/{exit}' /tmp/l3_all.s | sort -u | head -8
grep -A30 "^_ZN4ldsp5lean317icpc_lean3_kernelILi512ELi7ELb0ELb1EEE.*:" /tmp/l3_all.s > /dev/null
awk '/^[ \t]+(v_|s_|ds_|global_|buffer_|scratch_)/{t=$1; n++; if (t ~ /^v_mov/) mv++; else if (t ~ /^v_cndmask/) cm++; else if (t ~ /^v_cmp/) cp++; if (t ~ /^v_/) v++; else if (t ~ /^ds_/) l++; else if (t ~ /^s_nop/) np++; else if (t ~ /^s_waitcnt/) wc++; else if (t ~ /^s_/) s++; else g++}
     END{printf "static: %d instructions; VALU %d (moves %d, selects %d, compares %d); SALU %d; s_nop %d; s_waitcnt %d; LDS %d; memory %d\n", n, v, mv, cm, cp, s, np, wc, l, g}' /tmp/l3_512.s
