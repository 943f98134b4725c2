#!/bin/bash
# A named A-B build of the split-precision unit with extra -D flags, linked against the in-tree objects of the other units:
#   tools/variant_build.sh dephase -DRCV_C2_DEPHASE=2400   ->   tools/probe/ab/lib_dephase.so
#   RCV_LIB_PATH=$PWD/tools/probe/ab/lib_dephase.so python tools/fcn_bench.py --dtype f16x3 ...
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/tools/probe/ab
python -m rcvpose_amd.build > /dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/rcvpose_amd/csrc "$@" -c $R/rcvpose_amd/csrc/fcn_x3.hip -o $R/tools/probe/ab/fcn_x3_$NAME.o || exit 1
B=$R/rcvpose_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probe/ab/lib_$NAME.so $B/core.o $B/accumulator.o $B/fcn.o $B/fcn_f16.o $R/tools/probe/ab/fcn_x3_$NAME.o $B/pose.o $B/icp.o || exit 1
rm -f $R/tools/probe/ab/fcn_x3_$NAME.o
echo $R/tools/probe/ab/lib_$NAME.so
