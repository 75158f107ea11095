"""int16 PCM -> float mono conversion alone (tuning aid): one launch group's worth (512 ten-second clips), mono.   python tools/ubench/pcm_convert.py"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfann_amd.engine import Engine
eng = Engine(json.load(open("configs/default.json")), 0, max_batch=64)
g = torch.Generator(device="cuda"); g.manual_seed(3)
pcm = torch.randint(-30000, 30000, (512 * 80000,), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
for _ in range(3): w = eng.pcm16_to_mono(pcm)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): w = eng.pcm16_to_mono(pcm)
torch.cuda.synchronize()
print("pcm16_to_mono %.1f us per %d samples, checksum %.6f" % ((time.perf_counter() - t) / 50 * 1e6, pcm.numel(), w.double().sum().item()))
