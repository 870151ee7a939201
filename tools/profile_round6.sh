#!/bin/bash
# GPU box: the judged artefacts of a round-6 stage into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   tools/build_skip_lib.sh            (here, before: the DIAG build for the ablation table travels with the snapshot under tools/lab/lib/)
#   gpurun --timeout 3300 -- 'bash tools/profile_round6.sh r06_x'
# = tools/profile_round5.sh (bench line, the driver's 20-step command, rocprofv3 --kernel-trace --stats of it, launch order, per-launch table, DW traffic from separate --pmc
#   passes, ablation with four chains in flight, issue budget, the split-bf16 tables, k_pw_x3t under the counters, clock probe)
# + round 6: the HBM traffic of one whole forward (two --pmc passes, tools/net_traffic.sh), the MFMA-only floor of config[2] beside the kernel (tools/mfma_floor.py), the positive
#   controls of the two hazards (tests/test_gpu_hazard_controls.py, counts printed), the tree's commit.
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round5.sh $TAG
bash tools/net_traffic.sh > /dev/null 2>&1
cp gpurun_out/net_traffic.json gpurun_out/${TAG}_net_traffic.json; cp gpurun_out/net_traffic.txt gpurun_out/${TAG}_net_traffic.txt
timeout 300 python tools/mfma_floor.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_mfma_floor.txt
timeout 600 python -m pytest tests/test_gpu_hazard_controls.py -m gpu -q -rxX -s 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_hazard_controls.txt
tail -3 gpurun_out/${TAG}_net_traffic.txt | cut -c1-300; cat gpurun_out/${TAG}_mfma_floor.txt | cut -c1-300
