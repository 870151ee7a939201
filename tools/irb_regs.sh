#!/bin/sh
# register / scratch usage of every fused-block kernel instantiation (compile only, no GPU needed)
cd "$(dirname "$0")/../ffcnn_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -c ffgpu_kernels.hip -o /tmp/irb_regs.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: _Z5k_irbI" | grep -E "Function Name|VGPRs:|ScratchSize" |
    sed -e 's/.*Function Name: //' -e 's/.*VGPRs: / vgpr /' -e 's/.*ScratchSize \[bytes\/lane\]: / scratch /' -e 's/ \[-Rpass.*//' | paste - - - | grep "Li8EE"
