#!/bin/bash
# tools/exp_build.sh <name> [extra hipcc flags...]: compile the CURRENT csrc/kernels.hip (only) with the extra flags into
# mpas-ocean.jl_amd/build_<name>/kernels.o and link it with the product's other objects into libmoka_hip_<name>.so,
# for interleaved A/B runs on one GPU box (MOKA_HIP_LIB=<path> selects the library; tools/ab_libs.sh).
set -e
NAME=$1; shift
D=$(dirname "$0")/../mpas-ocean.jl_amd
make -C $D -j8 > /dev/null
mkdir -p $D/build_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result "$@" --offload-arch=gfx950 -c $D/csrc/kernels.hip -o $D/build_$NAME/kernels.o
OBJS=$(ls $D/build/*.o | grep -v "/kernels.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmoka_hip_$NAME.so $OBJS $D/build_$NAME/kernels.o
echo built $D/libmoka_hip_$NAME.so
