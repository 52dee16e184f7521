"""Step time and contact statistics as a function of the step index after a synchronous reset (the driver's bench run measures steps 5..25)."""
import importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
env = usim.UltrasoundVecEnv(4096, torso="soft", **usim.default_robosuite_kwargs())
blk = env.alloc_block(20)
env.reset_tensor(); torch.cuda.synchronize()
step = 0
for w in range(40):
    ms = env.time_steps(step, 20, blk); step += 20
    a = env.random_actions_tensor(step); env.step_tensor(a); step += 1
    nc = env.contacts[:, 0].float()
    ncmax_wave = env.contacts[:, 0].view(-1, 4).max(1).values.float()
    if w < 12 or w % 4 == 0:
        print(f"steps {step - 21:4d}-{step - 1:4d}: {ms / 20 * 1e3:6.2f} us/step   contacts mean {nc.mean():.2f}  per-wave max mean {ncmax_wave.mean():.2f}  envs with 8: {(nc == 8).float().mean():.3f}", flush=True)
