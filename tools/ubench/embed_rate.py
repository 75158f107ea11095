"""Embedding throughput (PCM in HBM -> fingerprints) of a config: python tools/ubench/embed_rate.py configs/n640d64.json"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfann_amd import lib as plib, synth
from pfann_amd.engine import Engine
cfgp = sys.argv[1] if len(sys.argv) > 1 else "configs/default.json"
params = json.load(open(cfgp))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 9728
eng = Engine(params, 0, max_batch=B)
eng.load_state_dict(synth.make_state_dict(params, seed=123))
pcm = synth.make_songs_torch(list(range(64)), 30.0, device="cuda").reshape(-1)
wav = eng.pcm16_to_mono(pcm)
starts = (torch.arange(B, device="cuda") * 1571) % (wav.shape[0] - 8000)
lib = plib.load()
for _ in range(2): e = eng.embed_windows(wav, starts)
lib.pfann_prof_reset(); lib.pfann_prof_enable(1)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): e = eng.embed_windows(wav, starts)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
lib.pfann_prof_enable(0)
buf = ctypes.create_string_buffer(4096); lib.pfann_prof_tags(buf, 4096)
print("%s: %.1f k segments/s (%.2f ms per %d)" % (cfgp, B / dt / 1e3, dt * 1e3, B))
for tag in buf.value.decode().split(","):
    c = ctypes.c_int64(0); ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(c))
    print("   %-26s %8.3f ms/step x%g" % (tag, ms / 3, c.value / 3))
