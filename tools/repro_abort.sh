#!/bin/bash
# Reproduce GPUTEST_r02's SIGABRT (pytest -m gpu died inside ffgpu_exec_forward_host): round 2's first 27 tests in round 2's order,
# N times in fresh processes, stderr not captured (-s), native backtrace on abort (tools/abtrace.c -> gpurun_out/repro_<tag>/trace.txt).
# usage: tools/repro_abort.sh <tag> <runs> [ENV=...]
tag=$1; runs=$2; shift 2
out=gpurun_out/repro_$tag; mkdir -p $out
gcc -shared -fPIC -O1 -o gpurun_out/abtrace.so tools/abtrace.c || exit 1
for i in $(seq 1 $runs); do
    env "$@" FFCNN_TEST_IN_CHILD=1 FFCNN_TEST_KEEP_ORDER=1 ABTRACE_OUT=$PWD/$out/trace.txt LD_PRELOAD=$PWD/gpurun_out/abtrace.so timeout 300 python -m pytest \
        tests/test_gpu_bench_modes.py tests/test_gpu_cfg_styles.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_api.py \
        -x -q -s -m gpu -p no:cacheprovider -p no:faulthandler > $out/run$i.log 2>&1
    rc=$?
    echo "$tag run $i rc=$rc" | tee -a $out/summary.txt
    if [ $rc -ne 0 ]; then echo "---- run $i" >> $out/trace.txt; tail -c 3000 $out/run$i.log > $out/fail$i.tail; else rm -f $out/run$i.log; fi
done
