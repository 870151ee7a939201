#!/bin/bash
# tools/loop_u8_split2.sh [runs] -- repeats the one GPU test that showed the LDS-DMA / concurrency slips of k_conv_x3 (two half-batch chains as parallel
# graph branches, records compared with the oracle and between two input paths) and counts failing runs
n=${1:-25}; f=0
for i in $(seq 1 $n); do
  python -m pytest tests/test_gpu_round3.py -q -x -p no:cacheprovider -k "test_u8_frames_into_the_first_kernel and 32-32" 2>&1 | grep -q "1 passed" || f=$((f+1))
done
echo "$f failures of $n ($FFCNN_HIP_LIB)"
