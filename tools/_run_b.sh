mkdir -p gpurun_out/r5c
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pw_x3t" 2>&1 | tail -3
python tools/pw_x3t_bench.py one > gpurun_out/r5c/x3t_bench2.txt 2>&1; cat gpurun_out/r5c/x3t_bench2.txt
