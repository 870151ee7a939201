#!/bin/bash
# GPU box: the judged artefacts of a round-2 stage into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/profile_round2.sh r02_a'
# bench line (default run), rocprofv3 --kernel-trace --stats of the driver's command (20 steps), one chain's launch order,
# per-launch table, the pointwise GEMM's counters (separate --pmc passes, no trace domains beside them).
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench20.json 2>> gpurun_out/${TAG}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_${TAG} -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${TAG}_fwd -o trace -- python $R/tools/run_forward.py 64 5 > /dev/null 2>&1)
python tools/trace_order.py $(find gpurun_out/prof_${TAG}_fwd -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_trace_order_b64.txt 2>&1
timeout 200 python tools/layer_profile.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_layers_b64.txt
timeout 300 tools/pmc.sh "k_pw_gemm32" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_LEVEL_VMEM TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" -- env FFGPU_PW_X3=0 FFGPU_PW_X3S=0 python $R/tools/pw_gemm_bench.py 2>&1 | grep "k_pw_gemm32" > gpurun_out/${TAG}_pw_gemm_pmc.txt
FFGPU_PW_X3=0 FFGPU_PW_X3S=0 timeout 120 python tools/pw_gemm_bench.py 2>&1 | grep pw_gemm >> gpurun_out/${TAG}_pw_gemm_pmc.txt
timeout 120 python tools/igemm_bench.py 2>&1 | grep igemm > gpurun_out/${TAG}_igemm_vs_generic.txt
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_fwd gpurun_out/pmc_tmp
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300
