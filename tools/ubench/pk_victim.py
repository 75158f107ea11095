"""Packed-fp32 arithmetic beside the batched fp16 scan (profiles/r5/NOTES.md section 6): the register-only kernel of
pk_victim.hip, packed (v_pk_fma_f32 / v_pk_mul_f32) and plain (v_fma_f32), on one stream while pfann_search_topk on 4085
query rows runs on another; output compared bit for bit with the quiet run.  Also beside a torch matmul (no fp16 MFMA scan)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
so = os.path.join(HERE, "libpk_victim.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-shared", "-fPIC",
                    os.path.join(HERE, "pk_victim.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.pk_victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
from pfann_amd.database import DeviceIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.nn.functional.normalize(torch.randn((200000, 128), device=dev, generator=g), dim=1)
q = torch.nn.functional.normalize(db[:4085] + 0.3 * torch.randn((4085, 128), device=dev, generator=g), dim=1).contiguous()
ix = DeviceIndex(128, 0); ix.load(db, np.array([0, 200000], np.int64), 0)
ix.search(q, 100)
N, ITERS = 256 * 256 * 16, int(os.environ.get("ITERS", "20000"))

def victim(packed):
    out = torch.empty(N, device=dev)
    assert lib.pk_victim_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(out.data_ptr()), N, ITERS, packed) == 0
    return out
ref = {p: victim(p).clone() for p in (1, 0)}
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for load in ("search", "matmul", "search"):
    for packed in (1, 0):
        bad = []
        for rep in range(8):
            with torch.cuda.stream(side):
                for _ in range(4):
                    if load == "search":
                        ix.search(q, 100)
                    else:
                        a = torch.randn(4096, 4096, device=dev); (a @ a).sum()
            got = victim(packed)
            torch.cuda.synchronize()
            bad.append(int((got != ref[packed]).sum()))
        print("beside", load, "| victim", "PACKED v_pk_*_f32" if packed else "plain v_fma_f32  ", "-> threads with a wrong result per repetition:", bad, flush=True)
