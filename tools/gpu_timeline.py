import importlib, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
# the phase stamps live in the profiling build of the library only
PROF = ROOT / "robotic-ultrasound-imaging_amd" / "lib" / "libusim_prof.so"
if not PROF.exists():
    subprocess.run(["make", "-C", str(ROOT / "robotic-ultrasound-imaging_amd" / "csrc"), "prof"], check=True)
os.environ["USIM_LIB"] = str(PROF)
sys.path.insert(0, str(ROOT))
import numpy as np, torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
# usage: python tools/gpu_timeline.py [soft|rigid[:lanes]] ...   (default: soft rigid)
cases = sys.argv[1:] or ["soft", "rigid"]
for case in cases:
    torso, _, lanes = case.partition(":")
    extra = {"lanes_per_env": int(lanes)} if lanes else {}
    env = usim.UltrasoundVecEnv(4096, torso=torso, **extra, **usim.default_robosuite_kwargs())
    env.reset_tensor(); env.rollout_random(0, 200); torch.cuda.synchronize()
    acc = {}
    for k in range(50):
        for name, v in env.profile_step(200 + k).items():
            acc.setdefault(name, []).append(v)
    tot = sum(np.median(v) for v in acc.values())
    print(f"== {case}: median shader-clock ticks per phase (wave 0 / workgroup 0), total {tot:.0f}")
    for name, v in acc.items():
        print(f"   {name:22s} {np.median(v):9.0f}  {100 * np.median(v) / tot:5.1f}%")
    env.close()
