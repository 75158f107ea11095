"""Timing of the sequence matcher on a worst-case candidate set (every top-k label in a different song)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfann_amd.database import DeviceIndex
n_songs, seg, d, k, nQ, ql = 16950, 59, 128, 100, 512, 19
n = n_songs * seg
g = torch.Generator(device="cuda"); g.manual_seed(1)
db = torch.randn((n, d), device="cuda", generator=g); db /= db.norm(dim=1, keepdim=True)
pos = np.arange(n_songs + 1, dtype=np.int64) * seg
idx = DeviceIndex(d, 0); idx.load(db, pos, 0)
src = (torch.arange(nQ, device="cuda") * 1931 + 7) % (n - 40)
rows = (src[:, None] + torch.arange(ql, device="cuda")[None, :]).reshape(-1)
q = db[rows] + 0.7 * torch.randn((nQ * ql, d), device="cuda", generator=g); q /= q.norm(dim=1, keepdim=True)
D, I = idx.search(q, k)
qs, qn = np.arange(nQ, dtype=np.int64) * ql, np.full(nQ, ql, np.int32)
for oo in (False, True):
    idx.song_lo = 0
    for _ in range(2): res, _ = idx.match(q, I, qs, qn, 1, 0.0, 0, oo, False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): res, _ = idx.match(q, I, qs, qn, 1, 0.0, 0, oo, False)
    torch.cuda.synchronize()
    print("only_owned", oo, "match ms", (time.perf_counter() - t) / 5 * 1e3, "mean n_cand", res["n_cand"].mean())
