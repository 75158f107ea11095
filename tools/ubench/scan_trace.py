"""Where a step of the batched full pass goes (tuning aid, round 6).  Needs a library built with -DPFANN_SCAN_TRACE:
    PFANN_HIPCC_FLAGS=-DPFANN_SCAN_TRACE python -m pfann_amd.build --force;  python tools/ubench/scan_trace.py [nq=9728]
Every wave of scan_f16_qres_kernel<8,false,64> sums the shader cycles of its step phases; printed: the mean per step."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfann_amd import lib as plib                              # noqa: E402
from pfann_amd.database import DeviceIndex                     # noqa: E402

d, n, k = 128, 1000000, 100
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 9728
g = torch.Generator(device="cuda")
g.manual_seed(5)
db = torch.randn((n, d), device="cuda", generator=g)
heads = db[::40].repeat_interleave(40, 0)[:n]
db = heads + 0.6 * db
db = db / db.norm(dim=1, keepdim=True)
q = torch.randn((nq, d), device="cuda", generator=g)
q[::2] = db[(torch.arange((nq + 1) // 2, device="cuda") * 7919) % n] + 0.5 * q[::2]
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
ix = DeviceIndex(d, 0)
ix.load(db.contiguous(), np.array([0, n], np.int64), 0)
lib = plib.load()
lib.pfann_debug_set_scan_trace.restype = ctypes.c_int
lib.pfann_debug_set_scan_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint]
for _ in range(2):
    ix.search(q, k)
cap = 1 << 14
buf = torch.zeros((cap, 4, 8), dtype=torch.int64, device="cuda")
assert lib.pfann_debug_set_scan_trace(buf.data_ptr(), cap) == 0
ix.search(q, k)
torch.cuda.synchronize()
assert lib.pfann_debug_set_scan_trace(None, 0) == 0
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 5] > 0]
steps = t[:, 5].astype(np.float64)
names = ["tile request", "fragment reads + MFMAs", "survivor epilogue", "wait for the next tile (vmcnt)", "barrier"]
print("%d waves traced, %.0f steps per wave; shader cycles per step (mean over waves, weighted by steps):" % (t.shape[0], steps.mean()))
tot = 0.0
for i, nm in enumerate(names):
    v = t[:, i].sum() / steps.sum()
    tot += v
    print("   %-34s %8.1f" % (nm, v))
print("   %-34s %8.1f" % ("sum", tot))
