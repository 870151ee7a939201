#!/bin/bash
# GPU box: regenerate the judged artefacts of a round into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
#   gpurun --timeout 1800 -- 'bash tools/profile_round.sh r01_f'
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_${TAG} -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${TAG}_fwd -o trace -- python $R/tools/run_forward.py 64 5 > /dev/null 2>&1)
python tools/trace_order.py $(find gpurun_out/prof_${TAG}_fwd -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_trace_order_b64.txt 2>&1
timeout 200 python tools/layer_profile.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_layers_b64.txt
timeout 600 python tools/tune_irbw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_irbw_sweep.txt
FFGPU_NO_IRBW=1 timeout 500 tools/irb_trace.sh > gpurun_out/${TAG}_irb_timeline_workgroup_kernel.txt 2>&1
timeout 500 tools/irb_trace.sh > gpurun_out/${TAG}_irb_timeline.txt 2>&1
timeout 400 bash tools/issue_budget.sh gpurun_out/${TAG}_issue_budget.txt > /dev/null 2>&1
timeout 400 python tools/ablate_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ablation_4streams.txt
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_fwd gpurun_out/pmc_issue
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
