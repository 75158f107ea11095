"""Exact top-k search by kernel tag at small and middle query-row counts (tuning aid): 1,000,000 unit rows x nq query rows,
k = 100, for nq in argv (default 19 76 152 304 608 1216 2432); labels checksum per nq so that two builds can be compared.
   python tools/ubench/scan_mid.py [nq ...]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfann_amd import lib as plib                              # noqa: E402
from pfann_amd.database import DeviceIndex                     # noqa: E402

d, n, k = 128, 1000000, 100
nqs = [int(a) for a in sys.argv[1:]] or [19, 76, 152, 304, 608, 1216, 2432]
g = torch.Generator(device="cuda")
g.manual_seed(5)
db = torch.randn((n, d), device="cuda", generator=g)
heads = db[::40].repeat_interleave(40, 0)[:n]
db = heads + 0.6 * db
db = db / db.norm(dim=1, keepdim=True)
qa = torch.randn((max(nqs), d), device="cuda", generator=g)
qa[::2] = db[(torch.arange((max(nqs) + 1) // 2, device="cuda") * 7919) % n] + 0.5 * qa[::2]
qa = (qa / qa.norm(dim=1, keepdim=True)).contiguous()
ix = DeviceIndex(d, 0)
ix.load(db.contiguous(), np.array([0, n], np.int64), 0)
lib = plib.load()
for nq in nqs:
    q = qa[:nq].contiguous()
    for _ in range(3):
        D, I = ix.search(q, k)
    torch.cuda.synchronize()
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        D, I = ix.search(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    lib.pfann_prof_reset(); lib.pfann_prof_enable(1)
    for _ in range(reps):
        D, I = ix.search(q, k)
    torch.cuda.synchronize()
    lib.pfann_prof_enable(0)
    buf = ctypes.create_string_buffer(4096); lib.pfann_prof_tags(buf, 4096)
    tags = []
    for tag in buf.value.decode().split(","):
        c = ctypes.c_int64(0); ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(c))
        if c.value:
            tags.append("%s %.1f us x%g" % (tag, 1e3 * ms / reps, c.value / reps))
    print("nq %5d: %8.1f us per call (%.2f us per row); labels checksum %d, score sum %.6f | %s"
          % (nq, dt * 1e6, dt * 1e6 / nq, int(I.sum().item()), float(D.double().sum().item()), "; ".join(tags)))
