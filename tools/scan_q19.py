#!/usr/bin/env python
"""Single-query scan regime: one 19-row query vs N unit-norm rows, exact top-100.
Reports the full-db pass' GB/s against the 8 TB/s HBM peak, from HIP events in the library."""
import ctypes
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pfann_amd import lib as plib
from pfann_amd.database import DeviceIndex


def main(n=1000050, d=128, nq=19, k=100, iters=30):
    dev = torch.device("cuda", 0)
    if os.environ.get("SCAN_PREALLOC_GB"):          # does what was allocated before the db change the pass time?
        dummy = torch.empty(int(float(os.environ["SCAN_PREALLOC_GB"]) * (1 << 30)), dtype=torch.uint8, device=dev)
        dummy.fill_(1)
    g = torch.Generator(device=dev); g.manual_seed(1)
    db = torch.randn((n, d), device=dev, generator=g); db /= db.norm(dim=1, keepdim=True)
    q = db[torch.arange(nq, device=dev) * 977 + 5] * 0.8 + 0.2 * torch.randn((nq, d), device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    idx = DeviceIndex(d, 0)
    idx.load(db, np.array([0, n], np.int64), 0)
    lib = plib.load()
    for _ in range(3):
        idx.search(q, k)
    lib.pfann_prof_reset(); lib.pfann_prof_enable(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        D, I = idx.search(q, k)
        if os.environ.get("SCAN_SYNC"):
            torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    lib.pfann_prof_enable(0)
    cnt = ctypes.c_int64(0)
    ms = lib.pfann_prof_elapsed_ms(b"scan_topk", ctypes.byref(cnt))
    us = 1e3 * ms / cnt.value
    out = {"n": n, "d": d, "nq": nq, "k": k, "scan_pass_us": round(us, 1),
           "db_GBps": round(n * d * 4 / us / 1e3, 1), "hbm_frac_of_8TBps": round(n * d * 4 / us / 1e3 / 8000, 4),
           "search_call_us": round(1e3 * e0.elapsed_time(e1) / iters, 1),
           "query_rows_per_s_scan_only": round(nq / (us * 1e-6), 0)}
    # exactness vs torch
    S = q @ db.T
    ref = torch.topk(S, k, dim=1).indices.sort(dim=1).values
    out["exact"] = bool(torch.equal(I.sort(dim=1).values, ref))
    print(json.dumps(out))
    assert out["exact"] or os.environ.get("SCAN_NOCHECK"), "top-k mismatch"


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
