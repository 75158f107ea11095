"""Is the encoder's output independent of what runs beside it?  Main stream: mono conversion + embed of 215 queries; a second
stream meanwhile: (A) nothing, (B) search + match of other fingerprints, (C) a torch matmul loop.  Bitwise comparison."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import gpu_workloads as gw
from pfann_amd import synth
from pfann_amd.database import DeviceIndex
params, sd, eng = gw.engine("default", 4096)
db, pos = gw.database(600, "default", 4096)
dev = eng.device
nq = 215
q_song = [int((j * 7919 + 13) % 600) for j in range(nq)]
qp, _ = synth.make_queries_torch(synth.make_songs_torch(q_song, 30.0, device=dev), list(range(nq)), 10.0, 0.0)
starts = (torch.arange(nq, device=dev)[:, None] * qp.shape[1] + torch.arange(19, device=dev)[None, :] * 4000).reshape(-1)
pcm = qp.reshape(-1).contiguous()
ix = DeviceIndex(128, 0); ix.load(db, pos, 0)
ref = eng.embed_windows(eng.pcm16_to_mono(pcm), starts).clone()
qs, ql = np.arange(nq, dtype=np.int64) * 19, np.full(nq, 19, np.int32)
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
D0, I0 = ix.search(ref, 100)
torch.cuda.synchronize()
for load in (os.environ.get("LOADS", "search,match,search_small,bound,search").split(",")):
    out = []
    for rep in range(10):
        if load is not None:
            with torch.cuda.stream(side):
                for _ in range(3):
                    if load == "search":
                        D, I = ix.search(ref, 100)
                    elif load == "match":
                        ix.match(ref, I0, qs, ql, 1, 0.0, 0, False, True, to_host=False)
                    elif load == "search_small":
                        for j in range(8):
                            ix.search(ref[j * 19:(j + 1) * 19].contiguous(), 100)
                    elif load == "bound":
                        lb = ix.reduce_bound(ix.search_bound(ref, 100, 58)[None], 100); ix.search_bounded(ref, 100, lb)
                    else:
                        a = torch.randn(4096, 4096, device=dev); (a @ a).sum()
        e = eng.embed_windows(eng.pcm16_to_mono(pcm), starts)
        torch.cuda.synchronize()
        bad = (e != ref).any(dim=1)
        out.append((int(bad.sum()), float((e - ref).abs().max())))
    print("beside the encoder:", load, "-> (windows differing, max abs diff) per repetition:", out, flush=True)
