"""In-process probe of the exchange-stream race: one RCCL rank, the sharded protocol forced, the exchange of a fixed batch
of fingerprints on the exchange stream while the caller's stream runs (A) the encoder, (B) a torch matmul, (C) nothing."""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29777")
os.environ["PFANN_EXCHANGE_STREAM"] = "1"
import gpu_workloads as gw
from pfann_amd import synth
from pfann_amd.database import DeviceIndex
from pfann_amd.dist import ShardedIndex
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
torch.cuda.set_device(0)
dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
params, sd, eng = gw.engine("default", 4096)
db, pos = gw.database(600, "default", 4096)
dev = eng.device
nq = 215
q_song = [int((j * 7919 + 13) % 600) for j in range(nq)]
qp, _ = synth.make_queries_torch(synth.make_songs_torch(q_song, 30.0, device=dev), list(range(nq)), 10.0, 0.0)
starts = (torch.arange(nq, device=dev)[:, None] * qp.shape[1] + torch.arange(19, device=dev)[None, :] * 4000).reshape(-1)
wav = eng.pcm16_to_mono(qp.reshape(-1))
emb = eng.embed_windows(wav, starts)
ix = DeviceIndex(128, 0); ix.load(db, pos, 0, song_range=(0, 600))
sh = ShardedIndex(ix, pos, 100, 1, 0.0, always_exchange=True)
assert sh.xs is not None
qs, ql = np.arange(nq, dtype=np.int64) * 19, np.full(nq, 19, np.int32)
torch.cuda.synchronize()
def once(load):
    emb_c = emb.clone()
    with sh.exchange(emb_c):
        D, I = sh.search_global(emb_c)
        res, ss = sh.match_global(emb_c, I, qs, ql, True, 0)
        done = torch.cuda.Event(); done.record()
    if load == "encoder":
        eng.embed_windows(wav, starts)
    elif load == "matmul":
        a = torch.randn(4096, 4096, device=dev); (a @ a).sum()
    done.synchronize(); torch.cuda.synchronize()
    return D.clone(), I.clone(), ss.clone(), emb_c
sh_xs = sh.xs
sh.xs = None
D0, I0, ss0, _ = once(None)
sh.xs = sh_xs
for load in (None, "matmul", "encoder", "encoder", None):
    bad = []
    for rep in range(12):
        D, I, ss, e = once(load)
        bad.append((int((ss != ss0).any(dim=2).sum()), int((I != I0).sum()), int((D != D0).sum()), int((e != emb).sum())))
    print(backend, "load", load, "-> (ss cells, labels, scores, emb elements) differing per repetition:", bad, flush=True)
dist.destroy_process_group()
