#!/bin/bash
# Dynamic instruction mix of one fused-block shape (rocprofv3 --pmc, counters only): bash tools/pmc_mix.sh ic ec oc stride HW res
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_CVT"; do
  rm -rf gpurun_out/pmc_mix
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_mix" -o t -- python "$R/tools/run_irb_only.py" "$@" > "$R/gpurun_out/pmc_mix.log" 2>&1 )
  DB=$(find gpurun_out/pmc_mix -name "*_results.db" | head -1)
  python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    if "irb" in k and "pack" not in k: print("%-40s %-28s %12.0f per launch" % (k.split("(")[0][:40], c, v / n))
PY
done
