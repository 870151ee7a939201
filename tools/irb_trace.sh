#!/bin/sh
# GPU box: build the tracing variant of the library, print the fused blocks' in-kernel timelines, restore the product build
cd "$(dirname "$0")/.." || exit 1
touch ffcnn_amd/csrc/ffgpu_kernels.hip && make -s -C ffcnn_amd/csrc TRACE=1 >/dev/null 2>&1
python tools/irb_trace.py 2>&1 | grep "trace" | cut -c1-330 | sed 's/ (us, mean.*//' 
touch ffcnn_amd/csrc/ffgpu_kernels.hip && make -s -C ffcnn_amd/csrc >/dev/null 2>&1
