#!/bin/bash
# HBM traffic of the depthwise kernel from PMC counters, one counter per pass (TCC slots: FETCH_SIZE 3,
# WRITE_SIZE 2 -- MI355X_MICROARCH.md), no tracing domains besides the implicit kernel dispatch records.
set -u
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o t -- python $GRAFT_REPO_ROOT/tools/run_dw_only.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/t_results.db gpurun_out/pmc_WRITE_SIZE/t_results.db | tee gpurun_out/dw3x3_traffic.txt
