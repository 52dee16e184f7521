import importlib, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
dmod = importlib.import_module("robotic-ultrasound-imaging_amd.distributed")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("pg ok", flush=True)
env = usim.UltrasoundVecEnv(4096, device=dev, **usim.default_robosuite_kwargs())
env.reset_tensor(); torch.cuda.synchronize(); print("reset ok", flush=True)
blk = env.alloc_block(128)
env.rollout_random(0, 128, blk); torch.cuda.synchronize(); print("rollout ok", flush=True)
g = dmod.RolloutGather(device=dev)
p = dmod.pack_block(blk); torch.cuda.synchronize(); print("pack ok", p.shape, flush=True)
out = g.gather(blk); torch.cuda.synchronize(); print("gather ok", out.shape, flush=True)
g.gather_async(blk); o = g.wait(); torch.cuda.synchronize(); print("async ok", o.shape, flush=True)
for it in range(6):
    env.rollout_random(128 * (it + 1), 128, blk); g.wait(); g.gather_async(blk)
g.wait(); torch.cuda.synchronize(); print("loop ok", flush=True)
dist.destroy_process_group()
