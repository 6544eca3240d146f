#!/bin/bash
# tools/dev_build.sh TAG [extra hipcc flags] — development library build/dev/libldsp_TAG.so: icpc_kernel.hip compiled with
# -DLDSP_DEV_512 (512-thread instantiations only, ~25 s) plus the given flags, linked with the production objects of the
# other translation units.  Select it with LDSP_HIP_LIB=build/dev/libldsp_TAG.so (tools/ and tests honour it).
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
CS=legenddsp.jl_amd/csrc
make -s -C $CS ldsp_api.o functor_kernels.o sipm_kernel.o
mkdir -p build/dev
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-variable -Wno-unused-function \
  -DLDSP_DEV_512 "$@" -c $CS/icpc_kernel.hip -o build/dev/icpc_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/libldsp_$TAG.so $CS/ldsp_api.o $CS/functor_kernels.o $CS/sipm_kernel.o build/dev/icpc_$TAG.o
echo build/dev/libldsp_$TAG.so
