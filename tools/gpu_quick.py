import importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
def bench(n=4096, steps=1000, torso="soft", **kw):
    env = usim.UltrasoundVecEnv(n, torso=torso, **kw, **usim.default_robosuite_kwargs())
    env.reset_tensor(); blk = env.alloc_block(128); env.rollout_random(0, 128, blk); torch.cuda.synchronize()
    ms = 0.0
    for b in range(steps // 128): ms += env.time_steps(128 * (b + 1), 128, blk)
    st = (steps // 128) * 128
    print(f"[{torso} {kw}] n={n} {ms / st * 1e3:.1f} us/step  {n * st / ms * 1e3:.3e} env-steps/s", flush=True)
    env.close()
for a in sys.argv[1:]:
    n, lpe = a.split(":")
    bench(int(n), lanes_per_env=int(lpe))
