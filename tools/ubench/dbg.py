import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfann_amd.database import DeviceIndex
n, d, nq, k = 200000, 128, 19, 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
db = torch.randn((n, d), device="cuda", generator=g); db /= db.norm(dim=1, keepdim=True)
q = db[torch.arange(nq, device="cuda") * 977 + 5] * 0.8 + 0.2 * torch.randn((nq, d), device="cuda", generator=g)
q /= q.norm(dim=1, keepdim=True)
idx = DeviceIndex(d, 0); idx.load(db, np.array([0, n], np.int64), 0)
D, I = idx.search(q, k)
S = q @ db.T
Dr, Ir = torch.topk(S, k, dim=1)
print("D err", (D - Dr).abs().max().item())
bad = (I.sort(1).values != Ir.sort(1).values).any(1)
print("bad rows", bad.nonzero().flatten().tolist())
r = 0
print(D[r, :8].tolist()); print(Dr[r, :8].tolist()); print(I[r,:8].tolist(), Ir[r,:8].tolist())
