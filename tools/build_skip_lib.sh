#!/bin/bash
# Tuning build of the library whose executor honours FFGPU_DBG_SKIP / FFGPU_DBG_KEEP (make DIAG=1: drops launches, results are
# wrong by design), built OUT OF TREE and placed under tools/lab/lib/libffcnn_hip_skip.so (never beside the product library: VERDICT r04) -- the product
# .so is never touched.  tools/ablate_layers.py / tools/profile_round5.sh load it through FFCNN_HIP_LIB + FFCNN_HIP_ALLOW_DIAG=1.  Run here (no GPU needed).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/skipbuild && mkdir -p /tmp/skipbuild/ffcnn_amd
cp -r $R/ffcnn_amd/csrc /tmp/skipbuild/ffcnn_amd/csrc && cp -r $R/include /tmp/skipbuild/include && mkdir -p /tmp/skipbuild/tools && cp $R/tools/isa_lint.py /tmp/skipbuild/tools/
rm -rf /tmp/skipbuild/ffcnn_amd/csrc/build
make -C /tmp/skipbuild/ffcnn_amd/csrc DIAG=1 ../lib/libffcnn_hip.so > /tmp/skipbuild/make.log 2>&1 || { tail -20 /tmp/skipbuild/make.log; exit 1; }
mkdir -p $R/tools/lab/lib
cp /tmp/skipbuild/ffcnn_amd/lib/libffcnn_hip.so $R/tools/lab/lib/libffcnn_hip_skip.so
ls -la $R/tools/lab/lib/libffcnn_hip_skip.so
