#!/bin/bash
# LDS pipe counters of one fused-block shape (rocprofv3 --pmc, counters only): bash tools/pmc_lds.sh ic ec oc stride HW res
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  rm -rf gpurun_out/pmc_mix
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_mix" -o t -- python "$R/tools/run_irb_only.py" "$@" > "$R/gpurun_out/pmc_mix.log" 2>&1 )
  DB=$(find gpurun_out/pmc_mix -name "*_results.db" | head -1)
  [ -z "$DB" ] && { tail -3 gpurun_out/pmc_mix.log; continue; }
  python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    if "irb" in k and "pack" not in k: print("%-40s %-28s %14.0f per launch" % (k.split("(")[0][:40], c, v / n))
PY
done
