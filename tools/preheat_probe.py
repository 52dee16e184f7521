"""Does the slow first block of bench.py come from the clock ramp of a GPU that has been idle?  First 20 timed steps (after reset + 5 steps)
without pre-heating, after 100 ms of torch matmuls, after 100 ms of elementwise traffic."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
mode = sys.argv[1]
dev = torch.device("cuda:0")
def heat(kind, ms=100):
    t0 = time.perf_counter()
    if kind == "mm":
        a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(5): a = (a @ b) * 1e-2
            torch.cuda.synchronize()
    elif kind == "ew":
        x = torch.randn(64 << 20, device=dev)
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(20): x.mul_(1.0001)
            torch.cuda.synchronize()
env = usim.UltrasoundVecEnv(4096, torso="soft", **usim.default_robosuite_kwargs())
blk = env.alloc_block(20)
env.reset_tensor()
heat(mode)
env.rollout_random(0, 5, blk); torch.cuda.synchronize()
out = []
for b in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout_random(5 + 20 * b, 20, blk); e1.record()
    out.append((e0, e1))
torch.cuda.synchronize()
print(mode, " ".join(f"{1e3 * a.elapsed_time(b) / 20:.1f}" for a, b in out))
