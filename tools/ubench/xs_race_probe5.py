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
wav0 = eng.pcm16_to_mono(qp.reshape(-1).contiguous()).clone()
segs = wav0[starts[:, None] + torch.arange(8000, device=dev)[None, :]]
segs = (segs - segs.mean(dim=1, keepdim=True)).contiguous()
mel0 = eng.melspec(segs).clone()
emb0 = eng.encode(mel0).clone()
ix = DeviceIndex(128, 0); ix.load(db, pos, 0)
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
# B: stray writes after the fact?
out1 = eng.melspec(segs); torch.cuda.synchronize()
with torch.cuda.stream(side):
    for _ in range(6): ix.search(emb0, 100)
torch.cuda.synchronize()
print("B: mel output changed by a LATER search:", int((out1 != mel0).sum()), "; segs changed:", int((segs != segs.clone()).sum()), flush=True)
# C: concurrency, with the pattern of what differs
for rep in range(6):
    with torch.cuda.stream(side):
        for _ in range(3): ix.search(emb0, 100)
    got = eng.melspec(segs)
    torch.cuda.synchronize()
    d = (got != mel0)
    rows = torch.nonzero(d.reshape(d.shape[0], -1).any(dim=1)).reshape(-1).tolist()
    desc = []
    for r in rows[:4]:
        dm = d[r]                                   # [256 mels, 32 frames]
        frames = torch.nonzero(dm.any(dim=0)).reshape(-1).tolist()
        mels = torch.nonzero(dm.any(dim=1)).reshape(-1)
        desc.append("win %d: frames %s, %d mel bins (first %s), max diff %.3g" % (r, frames, mels.numel(), mels[:6].tolist(), float((got[r] - mel0[r]).abs().max())))
    print("C rep", rep, "windows", len(rows), "|", " || ".join(desc), flush=True)
# D: same with the search replaced by a kernel that only occupies LDS-heavy CUs? (scan alone vs select alone cannot be split here)
