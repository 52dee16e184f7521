"""Development aid (run by hand on the GPU box): three-way comparison GPU / float64 oracle / float32 oracle around the first large force
difference of a full-size rollout.  Not collected by pytest."""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle_lib import Oracle
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
np.set_printoptions(precision=5, suppress=True, linewidth=220)
n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 130
env = usim.UltrasoundVecEnv(n, seed=3, torso="soft", **usim.default_robosuite_kwargs())
o64, o32 = Oracle(n, precision="f64", omp=True), Oracle(n, precision="f32")
env.reset(); o64.reset(); o32.reset()
alive = np.ones(n, bool)
for k in range(steps):
    a = o64.random_actions(k)
    r64, r32 = o64.step(a), o32.step(a)
    og, rg, dg, _ = env.step(a.astype(np.float32))
    cg = env.contacts.cpu().numpy()
    alive &= ~((dg != r64[2]) | (cg != r64[4]).any(1))
    d = np.abs(og - r64[0])[:, :3].max(1) * alive
    d32 = np.abs(r32[0] - r64[0])[:, :3].max(1) * alive * (r32[2] == r64[2]) * (r32[4] == r64[4]).all(1)
    bad = np.nonzero(d > 0.1)[0]
    if k % 10 == 0 or len(bad):
        print(f"step {k}: alive {alive.sum()} max dF gpu {d.max():.4f} (env {d.argmax()})  f32 oracle {d32.max():.4f} (env {d32.argmax()})")
    for i in bad[:4]:
        inf = o64.last_info()
        print(f"   env {i}: t {int(o64.get_state()['t'][i])} contacts {r64[4][i]}  margin {inf['contact_margin'][i]:.2e}  F64 {r64[0][i,:6]}  Fgpu {og[i,:6]}  F32 {r32[0][i,:6]}")
        alive[i] = False
