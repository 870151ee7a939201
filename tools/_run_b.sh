python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "front_kernel_columns" 2>&1 | tail -6
