#!/bin/bash
# tools/dev_build.sh TAG [extra hipcc flags] — development library build/dev/libldsp_TAG.so: icpc_lean3.hip compiled with
# -DLDSP_DEV_512 (512-thread instantiations only, ~20 s) plus the given flags, linked with cached -DLDSP_DEV_512 objects of
# icpc_lean.hip (config 2) / icpc_kernel.hip (rebuilt when their sources are newer) and the production objects of the other translation
# units.  Select it with LDSP_HIP_LIB=build/dev/libldsp_TAG.so (tools/ and tests honour it).
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
CS=legenddsp.jl_amd/csrc
make -s -C $CS ldsp_api.o functor_kernels.o sipm_kernel.o
mkdir -p build/dev
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function -DLDSP_DEV_512 -mllvm -amdgpu-atomic-optimizer-strategy=None"
newer() { [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ $CS/icpc_dev.hpp -nt "$2" ] || [ $CS/ldsp_device.hpp -nt "$2" ] || [ $CS/wave_prims.hpp -nt "$2" ]; }
if [ "${LDSP_DEV_GENERIC:-0}" = 1 ] || newer $CS/icpc_kernel.hip build/dev/icpc_generic.o; then
  /opt/rocm/bin/hipcc $FLAGS -c $CS/icpc_kernel.hip -o build/dev/icpc_generic.o &
fi
if newer $CS/icpc_lean.hip build/dev/icpc_lean2.o; then
  /opt/rocm/bin/hipcc $FLAGS -c $CS/icpc_lean.hip -o build/dev/icpc_lean2.o &
fi
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/icpc_lean3.hip -o build/dev/lean3_$TAG.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/libldsp_$TAG.so $CS/ldsp_api.o $CS/functor_kernels.o $CS/sipm_kernel.o build/dev/icpc_generic.o build/dev/icpc_lean2.o build/dev/lean3_$TAG.o
echo build/dev/libldsp_$TAG.so
