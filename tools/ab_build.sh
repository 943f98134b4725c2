#!/bin/bash
# Second build of the split-precision unit with extra -D flags, linked into tools/probe/ab/librcvpose_hip_a.so for A-B runs on one box:
#   tools/ab_build.sh -DRCV_X3_EARLY_LGKM=1 ;  RCV_LIB_PATH=$PWD/tools/probe/ab/librcvpose_hip_a.so python tools/fcn_bench.py ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/probe/ab
python -m rcvpose_amd.build > /dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/rcvpose_amd/csrc "$@" -c $R/rcvpose_amd/csrc/fcn_x3.hip -o $R/tools/probe/ab/fcn_x3_a.o || exit 1
B=$R/rcvpose_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probe/ab/librcvpose_hip_a.so $B/core.o $B/accumulator.o $B/fcn.o $B/fcn_f16.o $R/tools/probe/ab/fcn_x3_a.o $B/pose.o $B/icp.o || exit 1
rm -f $R/tools/probe/ab/fcn_x3_a.o
ls -la $R/tools/probe/ab/
