"""Where one 10 s query's 0.9 ms goes (tuning aid): wall time of each stage with a synchronise per call,
against a 1,000,050-row filler db.   python tools/ubench/one_query.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfann_amd import synth                                    # noqa: E402
from pfann_amd.database import DeviceIndex                     # noqa: E402
from pfann_amd.engine import Engine                            # noqa: E402
from pfann_amd.utils import read_config                        # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
params = read_config(os.path.join(REPO, "configs", "default.json"))
d, k = params["model"]["d"], params["indexer"]["top_k"]
dev = torch.device("cuda", 0)
eng = Engine(params, 0, max_batch=64)
eng.load_state_dict(synth.make_state_dict_calibrated(params, seed=123))
n_songs = 16950
song_pos = np.arange(n_songs + 1, dtype=np.int64) * 59
g = torch.Generator(device=dev)
g.manual_seed(1)
db = torch.randn((int(song_pos[-1]), d), device=dev, generator=g)
db = db / db.norm(dim=1, keepdim=True)
index = DeviceIndex(d, 0)
index.load(db, song_pos, 0)
pcm = synth.make_songs_torch([5], 30.0, device=dev)[0, :80000].contiguous()
starts = torch.arange(19, device=dev, dtype=torch.int64) * 4000
qs, ql = np.zeros(1, np.int64), np.full(1, 19, np.int32)


def timeit(f, n=200):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / n


emb = eng.embed_windows(eng.pcm16_to_mono(pcm), starts)
D, I = index.search(emb, k)
print("embed (mono + mel + encoder, 19 segments)  %7.1f us" % timeit(lambda: eng.embed_windows(eng.pcm16_to_mono(pcm), starts)))
print("search top-%d                              %7.1f us" % (k, timeit(lambda: index.search(emb, k))))
print("match, result to host                       %7.1f us" % timeit(lambda: index.match(emb, I, qs, ql)))
print("match, result left on the device            %7.1f us" % timeit(lambda: index.match(emb, I, qs, ql, to_host=False)))


def one():
    e = eng.embed_windows(eng.pcm16_to_mono(pcm), starts)
    D1, I1 = index.search(e, k)
    return index.match(e, I1, qs, ql)[0]


print("whole query                                 %7.1f us" % timeit(one))

# per-kernel-tag GPU time of the embed stage (HIP events around every launch; adds launch gaps of its own)
import ctypes                                                  # noqa: E402
from pfann_amd import lib as plib                              # noqa: E402
lib = plib.load()
lib.pfann_prof_reset()
lib.pfann_prof_enable(1)
for _ in range(50):
    one()
torch.cuda.synchronize()
lib.pfann_prof_enable(0)
buf = ctypes.create_string_buffer(8192)
lib.pfann_prof_tags(buf, 8192)
tot = 0.0
for tag in buf.value.decode().split(","):
    if tag:
        c = ctypes.c_int64(0)
        ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(c))
        tot += ms
        print("  %-28s %7.1f us per query  (%d launches)" % (tag, 1e3 * ms / 50, c.value // 50))
print("  sum of kernel times          %7.1f us per query" % (1e3 * tot / 50))
