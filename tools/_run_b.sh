bash tools/ab_bench.sh 2 400 libffcnn_hip.so libffcnn_hip.so:FFGPU_BRANCH=1 2>&1 | grep -v amdgpu.ids | tail -3
