"""Is the batched fp16 search a bad neighbour for OTHER kernels' LDS (torch ops that stage through shared memory), or is the
mel kernel the only victim?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import gpu_workloads as gw
from pfann_amd.database import DeviceIndex
params, sd, eng = gw.engine("default", 4096)
db, pos = gw.database(600, "default", 4096)
dev = eng.device
ix = DeviceIndex(128, 0); ix.load(db, pos, 0)
q = db[:4085].contiguous()
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(4096, 2048, device=dev, generator=g)
ops = {"sort": lambda: torch.sort(x, dim=1).values, "cumsum": lambda: torch.cumsum(x, dim=1), "softmax": lambda: torch.softmax(x, dim=1),
       "layer_norm": lambda: torch.nn.functional.layer_norm(x, (2048,)), "topk": lambda: torch.topk(x, 64, dim=1).values,
       "fft": lambda: torch.view_as_real(torch.fft.rfft(x, dim=1))}
ref = {k: f().clone() for k, f in ops.items()}
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for name, f in ops.items():
    bad = []
    for rep in range(8):
        with torch.cuda.stream(side):
            for _ in range(3): ix.search(q, 100)
        got = f()
        torch.cuda.synchronize()
        bad.append(int((got != ref[name]).sum()))
    print("victim", name, "-> elements differing per repetition:", bad, flush=True)
