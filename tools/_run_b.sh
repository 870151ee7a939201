mkdir -p gpurun_out/r5d
bash tools/ab_bench.sh 3 400 libffcnn_hip.so libffcnn_hip.so:FFGPU_FRONT_BAND=20 libffcnn_hip.so:FFGPU_FRONT_BAND=32 libffcnn_hip.so:FFGPU_THIN_BAND=20 libffcnn_hip.so:FFGPU_FRONT_BAND=20,FFGPU_THIN_BAND=20 libffcnn_hip.so:FFGPU_FRONT_BAND=8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5d/ab_bands.txt | tail -8
