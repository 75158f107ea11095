"""What the MAX-reduced bound buys a shard (tuning aid): 1,000,000 unit rows in songs of 40 similar rows, cut into
N shards held by N handles on this one GPU; time of shard 0's search for 9728 query rows, complete (pfann_search_topk) vs
two-phase (pfann_search_bound, k-th largest of the union of all shards' values, pfann_search_topk_bounded).   python tools/ubench/two_phase_search.py [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfann_amd.database import DeviceIndex                     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d, n, nq, k = 128, 1000000, 9728, 100
g = torch.Generator(device="cuda")
g.manual_seed(5)
db = torch.randn((n, d), device="cuda", generator=g)
heads = db[::40].repeat_interleave(40, 0)[:n]
db = heads + 0.6 * db
db = db / db.norm(dim=1, keepdim=True)
q = torch.randn((nq, d), device="cuda", generator=g)
q[::2] = db[(torch.arange(nq // 2, device="cuda") * 7919) % n] + 0.5 * q[::2]
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
cut = [n * i // N for i in range(N + 1)]
shards = []
for lo, hi in zip(cut[:-1], cut[1:]):
    ix = DeviceIndex(d, 0)
    ix.load(db[lo:hi].contiguous(), np.array([lo, hi], np.int64), lo)
    shards.append(ix)


def timeit(f, reps=10):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


m = min(k, 2 * k // N + 8)
L = shards[0].reduce_bound(torch.stack([ix.search_bound(q, k, m) for ix in shards]), k)
s0 = shards[0]


def two_phase():
    s0.search_bound(q, k, m)
    return s0.search_bounded(q, k, L)


print("%d shards of %d rows, %d query rows: complete local top-%d %.3f ms, two-phase with the reduced bound %.3f ms"
      % (N, cut[1], nq, k, timeit(lambda: s0.search(q, k)), timeit(two_phase)))
