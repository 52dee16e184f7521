"""Long soak through the multi-step launches (up to 256 steps per launch, state resident in registers between them): every randomisation on, random actions, checks per
256-step block: non-finite outputs, status bits, episode statistics.   usage: python tools/gpu_soak_blocks.py [n_envs] [steps] [mode] [control_freq]"""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
mode = sys.argv[3] if len(sys.argv) > 3 else "tracking"
freq = float(sys.argv[4]) if len(sys.argv) > 4 else 500.0
kw = usim.default_robosuite_kwargs()
kw["controller_configs"] = dict(kw["controller_configs"], impedance_mode=mode)
kw.update(deterministic_trajectory=False, torso_solref_randomization=True, initial_probe_pos_randomization=True, control_freq=freq)
env = usim.UltrasoundVecEnv(n, torso="soft", friction_randomization=True, seed=20211002, **kw)
env.reset_tensor()
T = 256
blk = env.alloc_block(T)
dev = env.device
bad = torch.zeros((), dtype=torch.int64, device=dev); dones = torch.zeros_like(bad); rsum = torch.zeros((), dtype=torch.float64, device=dev)
fault = torch.zeros_like(bad)
t0 = time.time()
for k in range(0, steps, T):
    env.rollout_random(k, T, blk)
    bad += (~torch.isfinite(blk["obs"])).sum() + (~torch.isfinite(blk["rew"])).sum()
    dones += blk["done"].sum(); rsum += blk["rew"].double().sum()
    fault += (env.status & 4).ne(0).sum()
torch.cuda.synchronize()
tot = n * (steps // T) * T
print(f"block soak mode={mode} control_freq={freq:g} envs={n} env-steps={tot} wall={time.time() - t0:.1f}s: episodes {int(dones)} (mean length {tot / max(int(dones), 1):.0f}), "
      f"reward/step {float(rsum) / tot:.3f}, non-finite outputs {int(bad)}, numerical-fault flags seen at block ends {int(fault)}")
env.close()
