"""ffcnn_amd -- MI355X-native (gfx950) implementation of rockcarry/ffcnn's forward
path behind ffcnn's own C API.

The product is ``ffcnn_amd/lib/libffcnn_hip.so`` (C host code + hand-written HIP
kernels, see ``include/*.h``).  This package is the thin Python mirror of that
C-ABI used by tests and bench.py: same function names, argument meaning and
error behaviour as the reference's ``ffcnn.h`` / ``conv.h``.  There is no CPU
fallback: importing works anywhere, but every compute call needs the built
library and a HIP device and fails loudly otherwise.
"""
from .capi import (BBOX, LAYER, NET, FrameDets, Executor, Net, FFGPU, build_library, groupconv, lib, library_path,
                   net_dump, net_forward, net_free, net_input, net_load)

__all__ = ["BBOX", "LAYER", "NET", "FrameDets", "Executor", "Net", "FFGPU", "build_library", "groupconv", "lib",
           "library_path", "net_dump", "net_forward", "net_free", "net_input", "net_load"]
