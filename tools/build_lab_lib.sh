#!/bin/bash
# A lab build of the library with extra compiler flags, OUT OF TREE, placed under tools/lab/lib/libffcnn_hip_<name>.so (never beside the product library):
#   tools/build_lab_lib.sh splitfirst -DXT_SPLIT_FIRST=1
# Load it with FFCNN_HIP_LIB=$PWD/tools/lab/lib/libffcnn_hip_<name>.so (tools/ab_bench.sh finds it by file name).  Run here (no GPU needed).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/labbuild_$NAME
rm -rf $B && mkdir -p $B/ffcnn_amd $B/tools
cp -r $R/ffcnn_amd/csrc $B/ffcnn_amd/csrc && cp -r $R/include $B/include && cp $R/tools/isa_lint.py $B/tools/
rm -rf $B/ffcnn_amd/csrc/build
make -C $B/ffcnn_amd/csrc EXTRA="$*" ../lib/libffcnn_hip.so > $B/make.log 2>&1 || { tail -20 $B/make.log; exit 1; }
mkdir -p $R/tools/lab/lib
cp $B/ffcnn_amd/lib/libffcnn_hip.so $R/tools/lab/lib/libffcnn_hip_$NAME.so
ls -la $R/tools/lab/lib/libffcnn_hip_$NAME.so
