FFGPU_PWXT_PERSIST=1 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pw_x3t" 2>&1 | tail -2
python tools/pw_x3t_bench.py one 2>&1 | grep -v amdgpu | cut -c1-500
