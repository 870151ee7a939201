import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ffcnn_amd import capi
net = capi.Net()
print("create", flush=True)
ex = net.executor(4, 32 | int(os.environ.get("XFLAGS", "0")))
print("created", ex.kernel_count, flush=True)
x = torch.rand((4, 3, 320, 320), device="cuda")
ex.forward_dev(x.data_ptr())
torch.cuda.synchronize()
print("forward ok", flush=True)
d = ex.read_dets()
print("dets", d["count"], flush=True)
ex.close()
print("closed", flush=True)
