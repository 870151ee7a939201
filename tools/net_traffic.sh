#!/bin/bash
# HBM traffic of one whole forward from PMC counters, one counter per pass (TCC: FETCH_SIZE, WRITE_SIZE -- MI355X_MICROARCH.md), counters only (no tracing domains):
#   gpurun -- 'bash tools/net_traffic.sh'   -> gpurun_out/net_traffic.json / .txt (copy to profiles/)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_net_$c
  timeout 900 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_net_$c -o t -- python $R/tools/net_traffic.py run > $R/gpurun_out/pmc_net_$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_net_$c.log
done
cd $R
python tools/net_traffic.py summary $(find gpurun_out/pmc_net_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/pmc_net_WRITE_SIZE -name "*_results.db" | head -1) | tee gpurun_out/net_traffic.txt
rm -rf gpurun_out/pmc_net_FETCH_SIZE gpurun_out/pmc_net_WRITE_SIZE
