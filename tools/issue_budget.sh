#!/bin/bash
# Issue budget of one 64-frame forward: vector-ALU and MFMA instruction counts per kernel (rocprofv3 --pmc pass on a few
# graph replays; counters only, no tracing).  bash tools/issue_budget.sh [out.txt]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/issue_budget.txt}
REPS=4
rm -rf gpurun_out/pmc_issue
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OLDPWD/gpurun_out/pmc_issue" -o t -- python "$OLDPWD/tools/run_graph.py" 64 $REPS 0 > "$OLDPWD/gpurun_out/pmc_issue.log" 2>&1 )
DB=$(find gpurun_out/pmc_issue -name "*_results.db" | head -1)
python tools/pmc_total.py "$DB" $REPS | tee "$OUT"
