#!/bin/bash
# Is a rare host crash a use-after-free?  glibc fills freed memory with a pattern (MALLOC_PERTURB_), which turns "rarely reads stale
# memory that happens to be intact" into "always reads junk".  Runs the lifecycle / leak tests under it, per variant, with a native
# backtrace on the fatal signal (tools/abtrace.c).  usage: tools/perturb_probe.sh <runs>
runs=${1:-2}
out=gpurun_out/perturb; mkdir -p $out
gcc -shared -fPIC -O1 -o gpurun_out/abtrace.so tools/abtrace.c || exit 1
cp ffcnn_amd/lib/libffcnn_hip.so $out/libffcnn_hip.so.snapshot
run() {
    tag=$1; shift
    for i in $(seq 1 $runs); do
        env "$@" MALLOC_PERTURB_=165 FFCNN_TEST_IN_CHILD=1 ABTRACE_OUT=$PWD/$out/trace_$tag.txt LD_PRELOAD=$PWD/gpurun_out/abtrace.so timeout 300 \
            python -m pytest tests/test_gpu_fuzz_api.py -x -q -s -m gpu -k "lifecycle or leak" -p no:cacheprovider -p no:faulthandler > $out/$tag.$i.log 2>&1
        echo "$tag run $i rc=$?" | tee -a $out/summary.txt
    done
}
run default
run nobranch FFGPU_BRANCH=0
run nograph FFGPU_NO_GRAPH=1
