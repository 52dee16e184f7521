"""step32 (two waves per quad of environments) against step16: bit-exactness and speed"""
import importlib, sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
def mk(n, lpe, **kw):
    return usim.UltrasoundVecEnv(n, device="cuda:0", seed=3, torso="soft", lanes_per_env=lpe, **kw, **usim.default_robosuite_kwargs())
a, b = mk(1000, 16), mk(1000, 32)
oa, ob = a.reset_tensor().clone(), b.reset_tensor().clone()
print("reset equal", torch.equal(oa, ob))
bad = 0
for k in range(300):
    act = a.random_actions_tensor(k).clone()
    ra = [x.clone() for x in a.step_tensor(act)]; rb = [x.clone() for x in b.step_tensor(act)]
    if not (torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]) and torch.equal(a.contacts, b.contacts)):
        bad += 1
        if bad < 3: print("step", k, "differs: obs max", float((ra[0] - rb[0]).abs().max()), "done", int((ra[2] != rb[2]).sum()))
print("steps with any difference:", bad, "of 300; episodes ended", int(sum(0 for _ in [])))
sa, sb = a.get_state(), b.get_state()
print("state equal", all(np.array_equal(sa[k], sb[k]) for k in sa))
a.close(); b.close()
for lpe in (16, 32, 32, 16):
    env = mk(4096, lpe)
    env.reset_tensor(); blk = env.alloc_block(128); env.rollout_random(0, 128, blk); torch.cuda.synchronize()
    ms = sum(env.time_steps(128 * (i + 1), 128, blk) for i in range(8))
    print(f"lanes {lpe}: {ms / 1024 * 1e3:.2f} us/step", flush=True)
    env.close()
