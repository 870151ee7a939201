#!/bin/bash
# tools/pmc.sh <kernel-substring> "<counter set 1>" ["<counter set 2>" ...] -- <command ...>
# One rocprofv3 --pmc pass per counter set (counters only: no trace domains beside them), per-launch averages of the
# kernels whose name contains the substring.  Run on the GPU box (gpurun).
cd "${GRAFT_REPO_ROOT:-$PWD}"; R=$PWD
K="$1"; shift
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
for set in "${SETS[@]}"; do
  rm -rf gpurun_out/pmc_tmp
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_tmp" -o t -- "$@" > "$R/gpurun_out/pmc_tmp.log" 2>&1 )
  DB=$(find gpurun_out/pmc_tmp -name "*_results.db" | head -1)
  [ -z "$DB" ] && { tail -5 gpurun_out/pmc_tmp.log; continue; }
  python - "$DB" "$K" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counters_collection" in x][0]
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from %s group by kernel_name, counter_name" % t).fetchall()
for k, c, v, n in rows:
    if sys.argv[2] in k: print("%-34s %-28s %16.0f per launch (%d launches)" % (k.split("(")[0][:34], c, v / n, n))
PY
done
