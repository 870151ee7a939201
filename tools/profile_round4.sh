#!/bin/bash
# GPU box: the judged artefacts of a round-4 stage into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   tools/build_skip_lib.sh            (here, before: the DIAG build for the ablation table travels with the snapshot)
#   gpurun --timeout 2400 -- 'bash tools/profile_round4.sh r04_x'
# = tools/profile_round3.sh (bench line, the driver's 20-step command, rocprofv3 --kernel-trace --stats of it, launch order, per-launch
# table, pointwise GEMM counters, igemm table, DW traffic from separate --pmc passes, ablation with four chains in flight)
# + the issue budget of one forward of the plan the bench runs (FFGPU_CONCURRENT): instruction counts and matrix-pipe busy cycles per
# kernel, counters only, one --pmc pass each
# + round 4's tables: split-bf16 fused blocks against the fp32-MFMA form, k_pw_x3 against the fp32 pointwise kernels, k_conv_x3 against the implicit
# GEMM (+ its counters), the other nets with and without it, DW variants.
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round3.sh $TAG
FLAGS=$(python -c "import sys; sys.path.insert(0, '$R'); from ffcnn_amd import capi; print(capi.FFGPU.CONCURRENT | capi.FFGPU.HOST_DETS)")
REPS=4
: > gpurun_out/${TAG}_issue_budget.txt
for CTRS in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    rm -rf gpurun_out/pmc_issue
    ( cd /tmp && timeout 600 rocprofv3 --pmc $CTRS -d "$R/gpurun_out/pmc_issue" -o t -- python "$R/tools/run_graph.py" 64 $REPS $FLAGS > "$R/gpurun_out/pmc_issue.log" 2>&1 )
    DB=$(find gpurun_out/pmc_issue -name "*_results.db" | head -1)
    [ -n "$DB" ] && python tools/pmc_total.py "$DB" $REPS >> gpurun_out/${TAG}_issue_budget.txt
done
rm -rf gpurun_out/pmc_issue
# the pointwise kernels of round 4 for config[2] under the counters: AUTO = the pointwise form of k_conv_x3 ("pw_x3s"), and k_pw_x3 (FFGPU_PW_X3S=0);
# profile_round2.sh's table above is k_pw_gemm32's (FFGPU_PW_X3=0 FFGPU_PW_X3S=0)
timeout 300 tools/pmc.sh "k_conv_x3<" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python $R/tools/pw_gemm_bench.py 2>&1 | grep "k_conv_x3<" > gpurun_out/${TAG}_pw_x3s_pmc.txt
timeout 120 python tools/pw_gemm_bench.py 2>&1 | grep "pw_" >> gpurun_out/${TAG}_pw_x3s_pmc.txt
timeout 300 python tools/pw_x3s_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pw_x3s.txt
export FFGPU_PW_X3S=0
timeout 300 tools/pmc.sh "k_pw_x3<" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum" -- python $R/tools/pw_gemm_bench.py 2>&1 | grep "k_pw_x3<" > gpurun_out/${TAG}_pw_x3_pmc.txt
timeout 120 python tools/pw_gemm_bench.py 2>&1 | grep "pw_" >> gpurun_out/${TAG}_pw_x3_pmc.txt
unset FFGPU_PW_X3S
timeout 300 python tools/other_nets.py --table 2>&1 | grep -v amdgpu.ids | cut -c1-1600 > gpurun_out/${TAG}_other_nets.txt
FFGPU_IG_X3=0 timeout 300 python tools/other_nets.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > gpurun_out/${TAG}_other_nets_fp32_igemm.txt
# dense 3x3 layers: k_conv_x3 against k_conv_igemm, and its counters on four layers (matrix-pipe busy cycles against SQ_BUSY_CU_CYCLES x 4 SIMDs)
timeout 300 python tools/conv_x3_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conv_x3_vs_igemm.txt
: > gpurun_out/${TAG}_conv_x3_pmc.txt
for SH in "384 256 64 26 26" "512 1024 64 13 13" "64 128 64 52 52" "16 32 64 208 208"; do
    timeout 120 python tools/conv_x3_one.py $SH 2>&1 | grep conv_x3 >> gpurun_out/${TAG}_conv_x3_pmc.txt
    timeout 300 tools/pmc.sh "k_conv_x3<" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python $R/tools/conv_x3_one.py $SH 2>&1 | grep "k_conv_x3<" >> gpurun_out/${TAG}_conv_x3_pmc.txt
done
timeout 300 python tools/x3_error.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_x3_fused_blocks.txt
timeout 300 python tools/pw_x3_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 > gpurun_out/${TAG}_pw_x3.txt
timeout 300 python tools/dw_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_dw_variants.txt
head -3 gpurun_out/${TAG}_issue_budget.txt
