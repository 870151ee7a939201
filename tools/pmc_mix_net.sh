#!/bin/bash
# Dynamic instruction mix of the whole 64-frame forward, per kernel (rocprofv3 --pmc passes, counters only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD; REPS=3
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"; do
  rm -rf gpurun_out/pmc_mix
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_mix" -o t -- python "$R/tools/run_graph.py" 64 $REPS 0 > "$R/gpurun_out/pmc_mix.log" 2>&1 )
  DB=$(find gpurun_out/pmc_mix -name "*_results.db" | head -1)
  python tools/pmc_total.py "$DB" $REPS
done
