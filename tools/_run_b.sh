mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_round5.py tests/test_gpu_bench_modes.py tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "group3 or bench or igemm or grouped or mini or tiny3" > gpurun_out/r5f/pytest.log 2>&1; tail -5 gpurun_out/r5f/pytest.log
python tools/other_nets.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 > gpurun_out/r5f/other_nets.txt; cat gpurun_out/r5f/other_nets.txt
