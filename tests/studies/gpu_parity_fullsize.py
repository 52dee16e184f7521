"""One-off record run (by hand, on the GPU box): the parity check of test_gpu_parity.py at BASELINE's full size -- 4096 environments x
200 steps -- for every controller mode and both torso models, plus the randomised configuration; prints the razor-edge counts.
Lives under tests/ because it uses the oracle.   python tests/studies/gpu_parity_fullsize.py > profiles/<round>/parity_fullsize.txt"""
import importlib, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
import test_gpu_parity as T
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
if len(sys.argv) > 1 and sys.argv[1] == "full":
    # the full torso (csrc/usim_full.h) at BASELINE's size against the oracle's full torso (dense algebra, OpenMP): 4096 environments x 200 steps per mode
    for mode in ("tracking", "fixed", "variable_z", "wrench"):
        t0 = time.time()
        explained, excluded = T._run_parity(usim, 4096, 200, "full", mode, omp=True)
        print(f"full  {mode:10s} 4096 envs x 200 steps: state within {T.STATE_RTOL:g} rel, body pose within 4e-6 m / 2e-5, done flags / contact indices bit-exact; "
              f"{explained} razor-edge decisions (probe-contact thresholds, table-contact onsets within float32 rounding of the plane), {excluded} environments excluded ({time.time() - t0:.0f} s)", flush=True)
    sys.exit(0)
for torso in ("rigid", "soft"):
    for mode in ("tracking", "fixed", "variable_z", "wrench"):
        t0 = time.time()
        explained, excluded = T._run_parity(usim, 4096, 200, torso, mode, omp=True)
        print(f"{torso:5s} {mode:10s} 4096 envs x 200 steps: state within {T.STATE_RTOL:g} rel, done flags / contact indices bit-exact; "
              f"{explained - T.LAST['f32_explained']} threshold decisions within rounding of the threshold in the oracle itself, {excluded} environments excluded; {T.LAST['f32_explained']} beyond the state bar (float32-explained) ({time.time() - t0:.0f} s)", flush=True)
for torso in ("rigid", "soft"):
    t0 = time.time()
    explained, excluded = T._run_parity(usim, 4096, 200, torso, "tracking", omp=True, robot="UR5e")
    print(f"{torso:5s} tracking   UR5e, 4096 envs x 200 steps: {explained - T.LAST['f32_explained']} razor-edge decisions, {excluded} environments excluded; {T.LAST['f32_explained']} environment(s) beyond the 1e-4 state bar, "
          f"explained by the oracle's own float32 build on the same field of the same environment ({time.time() - t0:.0f} s)", flush=True)
t0 = time.time()
explained, excluded = T._run_parity(usim, 8192, 200, "soft", "tracking", omp=True, friction_randomization=1, elem_friction=0.0, probe_friction=0.3)
print(f"soft  tracking   randomised friction/stiffness/damping (configs[4]), 8192 x 200: {explained - T.LAST['f32_explained']} razor-edge decisions, {excluded} environments excluded; {T.LAST['f32_explained']} beyond the state bar (float32-explained) ({time.time() - t0:.0f} s)")
