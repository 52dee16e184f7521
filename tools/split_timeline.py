"""Timeline of the split (two-wave) soft-torso step: shader-clock stamps of the arm wave and the lattice wave of workgroup 0 at their barriers."""
import importlib, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
PROF = ROOT / "robotic-ultrasound-imaging_amd" / "lib" / "libusim_prof.so"
subprocess.run(["make", "-s", "-C", str(ROOT / "robotic-ultrasound-imaging_amd" / "csrc"), "prof"], check=True)
os.environ["USIM_LIB"] = str(PROF)
sys.path.insert(0, str(ROOT))
import numpy as np, torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 200          # steps since the synchronous reset at which the stamps are taken
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 32          # 32: 16-lane groups, 64: 8-lane groups
env = usim.UltrasoundVecEnv(nenv, torso="soft", lanes_per_env=lanes, **usim.default_robosuite_kwargs())
env.reset_tensor(); env.rollout_random(0, pre); torch.cuda.synchronize()
rows = np.array([env.profile_step_raw(pre + k) for k in range(20 if pre < 100 else 50)], dtype=np.float64)
print(f"stamps of steps {pre} .. {pre + len(rows) - 1} after the reset")
t0 = rows[:, 20:21]
names_a = ["start", "fk done", "barrier1 passed", "op-space + controller done", "barrier2 passed", "precompute done", "barrier3 passed (contacts solved)", "finish done", "barrier4 passed", "end"]
names_b = ["start", "stage+rhs done", "barrier1 passed", "solve+collide done", "barrier2 passed", "contact solve done", "barrier3 passed", "integrate done", "barrier4 passed"]
for base, names, title in ((20, names_a, "arm wave"), (30, names_b, "lattice wave")):
    print("==", title)
    for i, nm in enumerate(names):
        v = rows[:, base + i] - t0[:, 0]
        print(f"   {nm:36s} {np.median(v):9.0f}")
print(f"   (contact_solve entered at {np.median(rows[:, 40] - t0[:, 0]):.0f}, left at {np.median(rows[:, 44] - t0[:, 0]):.0f}: median ticks since the step's start; ncmax of this quad varies)")
for i, nm in enumerate(["list merged", "ncmax known", "before contact_solve", "after contact_solve"]):
    print(f"   x{i} {nm:32s} {np.median(rows[:, 50 + i] - t0[:, 0]):9.0f}")
print("== contact solve of the lattice wave (ticks since its start)")
c0 = rows[:, 40:41]
for i, nm in enumerate(["start", "contact rows done", "Delassus blocks done", "sweeps done", "wrench done"]):
    print(f"   {nm:36s} {np.median(rows[:, 40 + i] - c0[:, 0]):9.0f}")
