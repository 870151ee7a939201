#!/bin/sh
# GPU box: per-phase timeline of a LONE wave (one frame) of the non-pipelined wave kernel; restores the product build
cd "$(dirname "$0")/.." || exit 1
touch ffcnn_amd/csrc/ffgpu_kernels.hip && make -s -C ffcnn_amd/csrc TRACE=1 >/dev/null 2>&1
FFGPU_IRBW_BIG=${BIG:-0} FFGPU_IRB_TRACE=1 python tools/irbw_lone.py 2>&1 | grep -E "trace|N=1:" | cut -c1-330 | sed 's/ (us, wave.*//'
touch ffcnn_amd/csrc/ffgpu_kernels.hip && make -s -C ffcnn_amd/csrc >/dev/null 2>&1
