"""What fixing the lattice rows' impedance at d_max leaves out (deviations): the oracle with MuJoCo's impedance ramp d(|r|) evaluated on every
joint-equality and tendon row (uso_config.lattice_ramp = 1: the lattice matrix is assembled and factorised per step) against the product's model, on the same
seeded episodes -- random actions, and the reference's trained `tracking` policy (weights / VecNormalize statistics from tests/golden).  Oracle only (CPU).
usage: python tests/studies/lattice_ramp_study.py [n_envs] [steps]  ->  profiles/r03/lattice_ramp_study.txt"""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from oracle_lib import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kw = dict(torso="top", seed=3, torso_solref_randomization=1, initial_probe_pos_randomization=1, early_termination=1)

def stats(tag, o0, o1, alive):
    f0, f1 = o0[alive, :3], o1[alive, :3]
    con = np.abs(f1[:, 2]) > 1e-9
    d = np.linalg.norm(f0 - f1, axis=1)[con]; fn = np.linalg.norm(f1, axis=1)[con]
    if con.sum() == 0:
        print(f"{tag}: no contacts"); return
    print(f"{tag}: {int(con.sum())} environments in contact; |F| median {np.median(fn):6.2f} N; |dF| median {np.median(d):.4f} N ({np.median(d / fn) * 100:.2f} %), "
          f"90 % {np.quantile(d, 0.9):.4f} N ({np.quantile(d / fn, 0.9) * 100:.2f} %)")

print(f"{n} environments, {steps} steps; a = impedance fixed at d_max (product), b = MuJoCo's ramp on every lattice row")
a, b = Oracle(n, **kw), Oracle(n, lattice_ramp=1, **kw)
oa, ob = a.reset(), b.reset()
stats("reset observation          ", oa, ob, np.ones(n, bool))
alive = np.ones(n, bool)
for k in range(steps):
    act = a.random_actions(k)
    oa, ra, da, _, _ = a.step(act, auto_reset=False); ob, rb, db, _, _ = b.step(act, auto_reset=False)
    alive &= ~(da.astype(bool) | db.astype(bool))
    if k + 1 in (1, 5, 20, 50, 100, 200, steps):
        stats(f"random actions, step {k + 1:4d}", oa, ob, alive)
sa, sb = a.get_state(), b.get_state()
print(f"  lattice displacement after {steps} steps (environments still running in both): max |s| {np.abs(sb['s'][alive]).max() * 1e3:.2f} mm, max |ds| {np.abs(sa['s'][alive] - sb['s'][alive]).max() * 1e3:.3f} mm")

# the reference's trained policy, evaluated as src/rl.py:171-192 does (VecNormalize statistics frozen), stochastic
meta = json.loads((ROOT / "tests/golden/reference_pins.json").read_text())["tracking"]
pins = np.load(ROOT / "tests/golden/reference_pins.npz")
W = {k: v.astype(np.float64) for k, v in np.load(ROOT / "tests/golden/tracking_policy.npz").items()}
mean, var = pins["tracking_obs_rms_mean"], pins["tracking_obs_rms_var"]
def policy(obs, rng):
    x = np.clip((obs - mean) / np.sqrt(var + meta["epsilon"]), -meta["clip_obs"], meta["clip_obs"])
    h = np.tanh(x @ W["mlp_extractor.policy_net.0.weight"].T + W["mlp_extractor.policy_net.0.bias"])
    h = np.tanh(h @ W["mlp_extractor.policy_net.2.weight"].T + W["mlp_extractor.policy_net.2.bias"])
    mu = h @ W["action_net.weight"].T + W["action_net.bias"]
    return np.clip(mu + np.exp(W["log_std"]) * rng.standard_normal(mu.shape), 0.0, 1.0)
T = max(steps, 600)
print(f"reference `tracking` policy, {n} environments x {T} steps each (auto-reset):")
for tag, ramp in (("a (d fixed at d_max)", 0), ("b (impedance ramp)  ", 1)):
    o = Oracle(n, lattice_ramp=ramp, **kw); rng = np.random.default_rng(0)
    obs = o.reset(); acc = []; rew = 0.0; ndone = 0
    for k in range(T):
        obs, r, d, _, _ = o.step(policy(obs, rng))
        acc.append(obs.copy()); rew += r.sum(); ndone += int(d.sum())
    acc = np.concatenate(acc)
    print(f"  {tag}: reward/step {rew / (n * T):.3f}, episodes ended {ndone}; contact force mean ({acc[:, 0].mean():6.2f}, {acc[:, 1].mean():6.2f}, {acc[:, 2].mean():6.2f}) N, "
          f"std ({acc[:, 0].std():5.2f}, {acc[:, 1].std():5.2f}, {acc[:, 2].std():5.2f}); height above the trajectory {acc[:, 14].mean() * 1e3:.2f} mm")
print(f"  MuJoCo (vec_normalize_tracking.pkl, 40 M steps): contact force mean ({mean[0]:6.2f}, {mean[1]:6.2f}, {mean[2]:6.2f}) N, std ({np.sqrt(var[0]):5.2f}, {np.sqrt(var[1]):5.2f}, "
      f"{np.sqrt(var[2]):5.2f}); height above the trajectory {mean[14] * 1e3:.2f} mm")
