"""melspec kernel alone on 9728 segments (timing / PMC runs)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfann_amd.engine import Engine
params = json.load(open("configs/default.json"))
eng = Engine(params, 0, max_batch=64)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((9728, 8000), device="cuda", generator=g) * 0.1
for _ in range(2): out = eng.melspec(x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): out = eng.melspec(x)
torch.cuda.synchronize()
print("melspec ms per 9728 segments: %.3f  (group %s)  checksum %.6f" % ((time.perf_counter() - t) / 5 * 1e3, os.environ.get("PFANN_MEL_GROUP", "auto"), out.double().sum().item()))
