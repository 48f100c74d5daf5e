#!/bin/bash
# Developer tool: build variants of libsmpc.so that differ in experiment switches of the lane pass
# (smpc_lane.hip LANE_X_*, smpc_device_math.h SMPC_X_*), side by side under
# mpcholonavigation_amd/variants/, for tools/kbench_all.py.   tools/variants.sh name "-DFLAG=0 ..." ...
set -e
cd "$(dirname "$0")/../mpcholonavigation_amd/csrc"
make -s
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -Wno-pass-failed $defs -c -o ../variants/lane_$name.o smpc_lane.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libsmpc_$name.so smpc_kernels.o ../variants/lane_$name.o smpc_split.o \
        smpc_api.o smpc_prepare.o smpc_shard.o smpc_group.o -Wl,-rpath,/opt/rocm/lib
    echo built $name ) &
done
wait
