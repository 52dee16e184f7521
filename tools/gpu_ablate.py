import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
def bench(n=4096, steps=300, torso="soft", **kw):
    env = usim.UltrasoundVecEnv(n, torso=torso, **kw, **usim.default_robosuite_kwargs())
    env.reset_tensor(); blk = env.alloc_block(steps); env.rollout_random(0, 100, None); torch.cuda.synchronize()
    ms = env.time_steps(100, steps, blk)
    print(f"[{torso} {kw}] n={n} {ms / steps * 1e3:.1f} us/step  {n * steps / ms * 1e3:.3e} env-steps/s", flush=True)
    env.close()
for lpe in (16, 8):
    for it in (0, 1, 2, 5, 10):
        bench(pgs_iters=it, lanes_per_env=lpe)
bench(torso="rigid", lanes_per_env=16); bench(torso="rigid", lanes_per_env=1)
bench(n=256, lanes_per_env=16); bench(n=256, torso="rigid", lanes_per_env=16); bench(n=16, lanes_per_env=16)
