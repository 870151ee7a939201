#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, kernel-trace profile.  Run via
#   gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > gpurun_out/lscpu.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps ${STEPS:-30} --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit: $?" >> gpurun_out/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_stats*" | head -3 | while read f; do head -40 "$f" > gpurun_out/kernel_stats_head.csv; done
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err
