#!/bin/bash
# GPU box: the judged artefacts of a round-3 stage into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   gpurun --timeout 1800 -- 'bash tools/profile_round3.sh r03_x'
# = tools/profile_round2.sh (bench line incl. c_node_api and u8 input, the driver's 20-step command, rocprofv3 --kernel-trace --stats
# of it, one chain's launch order, per-launch table, the pointwise GEMM's counters, igemm table) + the DW kernel's HBM traffic from
# separate --pmc passes + the marginal cost of each stretch of the net with four chains in flight (DIAG build of the library in /tmp).
set -u
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
bash tools/profile_round2.sh $TAG
bash tools/pmc_traffic.sh > /dev/null 2>&1
cp gpurun_out/dw3x3_traffic.txt gpurun_out/${TAG}_dw3x3_traffic.txt 2>/dev/null
cp gpurun_out/dw3x3_traffic.json gpurun_out/${TAG}_dw3x3_traffic.json 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
# ablation: the DIAG build of the library (tools/build_skip_lib.sh, built beforehand where the sources are: it travels with the snapshot)
if [ -f tools/lab/lib/libffcnn_hip_skip.so ]; then
    FFCNN_HIP_ALLOW_DIAG=1 FFCNN_HIP_LIB=$R/tools/lab/lib/libffcnn_hip_skip.so timeout 600 python tools/ablate_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ablation_4streams.txt
    tail -3 gpurun_out/${TAG}_ablation_4streams.txt
fi
