mkdir -p gpurun_out/r5h
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_fuzz_nets.py tests/test_gpu_round5.py -m gpu -x -q -k "irb or block or fused or every_layer or random_nets or geometry or concurrent" > gpurun_out/r5h/pytest2.log 2>&1; tail -4 gpurun_out/r5h/pytest2.log
bash tools/ab_bench.sh 3 400 libffcnn_hip_noisel.so libffcnn_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5h/ab_isel.txt | tail -3
