"""What the eight contact slots leave out (deviations): the oracle built with 32 slots (every penetrating element keeps its contact, as
in MuJoCo) against the regular build on the same seeded resets and random-action steps.  Oracle only (CPU).
usage: python tests/studies/slot_overflow_study.py [n_envs] [steps]  ->  profiles/r03/slot_overflow_study.txt"""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle_lib import Oracle, OracleConfig, ORACLE_DIR

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "allcontacts"], check=True)
ref = Oracle(n, torso="top", seed=3, torso_solref_randomization=1, initial_probe_pos_randomization=1)
lib = C.CDLL(str(ORACLE_DIR / "_build" / "libusim_oracle_f64_allcontacts.so"))
lib.uso_create.restype = C.c_void_p; lib.uso_create.argtypes = [C.POINTER(OracleConfig), C.c_int]
dp = C.POINTER(C.c_double)
lib.uso_reset.argtypes = [C.c_void_p, C.c_void_p, dp]
lib.uso_step.argtypes = [C.c_void_p, dp, dp, dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.uso_last_info.argtypes = [C.c_void_p, dp]
h = lib.uso_create(C.byref(ref.cfg), n)
obs_all = np.zeros((n, 19)); obs8 = ref.reset()
lib.uso_reset(h, None, obs_all.ctypes.data_as(dp))
info = np.zeros((n, 8)); lib.uso_last_info(h, info.ctypes.data_as(dp))
st = ref.get_state()["status"].astype(int)
over = (st & 1) != 0
def report(tag, a, b, mask):
    f8, fa = a[mask, :3], b[mask, :3]
    dn = np.linalg.norm(f8 - fa, axis=1); fn = np.maximum(np.linalg.norm(fa, axis=1), 1e-9)
    tq = np.linalg.norm(a[mask, 3:6] - b[mask, 3:6], axis=1)
    print(f"{tag}: {int(mask.sum())} environments; contact force |dF| median {np.median(dn):.3f} N, 90 % {np.quantile(dn, 0.9):.3f} N, max {dn.max():.2f} N; "
          f"relative to |F| median {np.median(dn / fn) * 100:.2f} %, 90 % {np.quantile(dn / fn, 0.9) * 100:.2f} %; |F| median {np.median(fn):.1f} N; torque sensor |d| median {np.median(tq):.4f} N m")
print(f"{n} seeded resets (position noise on: the probe is spawned up to 3 cm deep, ultrasound.py:880): {int(over.sum())} ({over.mean() * 100:.1f} %) start with more than 8 penetrating elements")
print("eight deepest contacts (product) vs every contact (MuJoCo keeps all):")
report("  reset observation, overflowing environments", obs8, obs_all, over)
report("  reset observation, the others              ", obs8, obs_all, ~over)
rew_d = []
alive = np.ones(n, bool)
for k in range(steps):
    a = ref.random_actions(k)
    o8, r8, d8, _, _ = ref.step(a, auto_reset=False)
    oa, ra, da = np.zeros((n, 19)), np.zeros(n), np.zeros(n, np.uint8)
    lib.uso_step(h, a.ctypes.data_as(dp), oa.ctypes.data_as(dp), ra.ctypes.data_as(dp), da.ctypes.data_as(C.c_void_p), None, None, 0)
    alive &= ~(d8.astype(bool) | da.astype(bool))
    if k in (0, 4, 9, steps - 1):
        report(f"  step {k + 1:3d}, environments that started overflowing", o8, oa, over & alive)
    rew_d.append(np.abs(r8 - ra)[over & alive].mean() if (over & alive).any() else 0.0)
print(f"  mean |d reward| over the first {steps} steps of those environments: {np.mean(rew_d):.4f} (reward per step ~ 6)")
