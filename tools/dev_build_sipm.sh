#!/bin/bash
# tools/dev_build_sipm.sh TAG [extra hipcc flags] — development library build/dev/libldsp_TAG.so with sipm_kernel.hip compiled with
# the given flags (e.g. -DLDSP_STAMPS), linked with the production objects of the other translation units.
# Select it with LDSP_HIP_LIB=build/dev/libldsp_TAG.so.
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
CS=legenddsp.jl_amd/csrc
make -s -C $CS -o sipm_s4.inc ldsp_api.o functor_kernels.o icpc_kernel.o icpc_lean.o icpc_lean3.o   # (sipm_s4.inc is included by sipm_kernel.hip only)
mkdir -p build/dev
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -c $CS/sipm_kernel.hip -o build/dev/sipm_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/libldsp_$TAG.so $CS/ldsp_api.o $CS/functor_kernels.o $CS/icpc_kernel.o $CS/icpc_lean.o $CS/icpc_lean3.o build/dev/sipm_$TAG.o
echo build/dev/libldsp_$TAG.so
