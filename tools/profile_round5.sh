#!/bin/bash
# GPU box: the judged artefacts of a round-5 stage into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   tools/build_skip_lib.sh            (here, before: the DIAG build for the ablation table travels with the snapshot under tools/lab/lib/)
#   gpurun --timeout 2700 -- 'bash tools/profile_round5.sh r05_x'
# = tools/profile_round4.sh (bench line, the driver's 20-step command, rocprofv3 --kernel-trace --stats of it, launch order, per-launch table, DW traffic
#   from separate --pmc passes, ablation with four chains in flight, issue budget, the split-bf16 tables of round 4)
# + round 5: BASELINE config[2]'s kernel (k_pw_x3t) against its predecessors and under the counters (counters only, one --pmc pass per set), the shader
#   clock the chip holds under the mix / config[1] / config[2] (tools/clock_mix.py), the tree's commit.
set -u
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round4.sh $TAG
timeout 300 python tools/pw_x3t_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pw_x3t.txt
FFGPU_PWX3T_MIN_IC=8 timeout 600 bash tools/pmc.sh "k_pw_x3t<" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
    "SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
    -- python $R/tools/pw_x3t_bench.py pmc 2>&1 | grep "k_pw_x3t<\|GFLOP" > gpurun_out/${TAG}_pw_x3t_pmc.txt
timeout 300 python tools/clock_mix.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_clock_mix.txt
cat gpurun_out/${TAG}_clock_mix.txt
