python -m pytest tests/test_gpu_round5.py -q -m gpu -k "group3" 2>&1 | tail -12
