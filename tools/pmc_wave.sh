cd "$GRAFT_REPO_ROOT"; R=$PWD
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  rm -rf gpurun_out/pmc_mix
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_mix" -o t -- python "$R/tools/run_irb_only.py" "$@" > "$R/gpurun_out/pmc_mix.log" 2>&1 )
  DB=$(find gpurun_out/pmc_mix -name "*_results.db" | head -1)
  [ -z "$DB" ] && { tail -3 gpurun_out/pmc_mix.log; continue; }
  python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    if "irb" in k and "pack" not in k: print("%-30s %-28s %14.0f per launch" % (k.split("(")[0][:30], c, v / n))
PY
done
