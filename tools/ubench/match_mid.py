"""Sequence matcher at small and middle query counts (tuning aid, round 6): ms per call and a checksum of the decisions for
nQ in argv (default 4 16 32 64 128 256), on the bench-like worst case (every top-k label in a different song).
   PFANN_MATCH_PHASED_MAX=n python tools/ubench/match_mid.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfann_amd.database import DeviceIndex
n_songs, seg, d, k, ql = 16950, 59, 128, 100, 19
nqs = [int(a) for a in sys.argv[1:]] or [4, 16, 32, 64, 128, 256]
n = n_songs * seg
g = torch.Generator(device="cuda"); g.manual_seed(1)
db = torch.randn((n, d), device="cuda", generator=g); db /= db.norm(dim=1, keepdim=True)
pos = np.arange(n_songs + 1, dtype=np.int64) * seg
idx = DeviceIndex(d, 0); idx.load(db, pos, 0)
for nQ in nqs:
    src = (torch.arange(nQ, device="cuda") * 1931 + 7) % (n - 40)
    rows = (src[:, None] + torch.arange(ql, device="cuda")[None, :]).reshape(-1)
    q = db[rows] + 0.7 * torch.randn((nQ * ql, d), device="cuda", generator=g); q /= q.norm(dim=1, keepdim=True)
    D, I = idx.search(q, k)
    qs, qn = np.arange(nQ, dtype=np.int64) * ql, np.full(nQ, ql, np.int32)
    for _ in range(3): res, _ = idx.match(q, I, qs, qn, 1, 0.0, 0, False, False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): res, _ = idx.match(q, I, qs, qn, 1, 0.0, 0, False, False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    chk = int(np.sum(res["song"].astype(np.int64) * 131 + res["offset"].astype(np.int64))), float(np.sum(res["score"].astype(np.float64)))
    print("nQ %4d: match %.4f ms per call; decisions checksum %d, score sum %.9f" % (nQ, ms, chk[0], chk[1]))
