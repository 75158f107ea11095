"""Batched exact top-k search by kernel tag (tuning aid): 1,000,000 unit rows in songs of 40 similar rows x 9728 query rows,
k = 100; also prints a checksum of the labels so that two builds can be compared.   python tools/ubench/scan_batched.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfann_amd import lib as plib                              # noqa: E402
from pfann_amd.database import DeviceIndex                     # noqa: E402

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
ix = DeviceIndex(d, 0)
ix.load(db.contiguous(), np.array([0, n], np.int64), 0)
lib = plib.load()
for _ in range(2):
    D, I = ix.search(q, k)
torch.cuda.synchronize()
reps = 10
lib.pfann_prof_reset(); lib.pfann_prof_enable(1)
t = time.perf_counter()
for _ in range(reps):
    D, I = ix.search(q, k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
lib.pfann_prof_enable(0)
buf = ctypes.create_string_buffer(4096); lib.pfann_prof_tags(buf, 4096)
print("search %d x %d, k = %d: %.3f ms per call; labels checksum %d, score sum %.6f"
      % (nq, n, k, dt * 1e3, int(I.sum().item()), float(D.double().sum().item())))
for tag in buf.value.decode().split(","):
    c = ctypes.c_int64(0); ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(c))
    if c.value:
        print("   %-26s %8.3f ms/call x%g" % (tag, ms / reps, c.value / reps))
