"""Sustained-load gap: the reference's trained policies hold their 5 N goal force 4-6 mm deeper here than on MuJoCo (end-of-training samples in
the reference's VecNormalize pickles).  Which single parameter of the restated model would have to change to close it?  The oracle replays the `wrench` policy
(force-controlled: the cleanest probe of the torso's sustained stiffness) and the `tracking` policy with the time constant of the lattice's joint-equality rows
varied (study switch uso_config.study_fix_tc; the instantaneous response -- the reset rows the probe stand-in is calibrated on -- does not depend on it).
Oracle only (CPU).   usage: python tests/studies/sustained_load_study.py [n_envs] [steps]"""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from oracle_lib import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 700
metas = json.loads((ROOT / "tests/golden/reference_pins.json").read_text())
pins = np.load(ROOT / "tests/golden/reference_pins.npz")
MODE = {"tracking": 0, "wrench": 3}
def med(o):
    o = o[o[:, 2] > 0]
    return (f"Fz median {np.median(o[:, 2]):5.2f} N, height above the trajectory {np.median(o[:, 14]) * 1e3:5.2f} mm ({np.quantile(o[:, 14], .25) * 1e3:5.2f} .. "
            f"{np.quantile(o[:, 14], .75) * 1e3:5.2f}), |Fx| {np.median(np.abs(o[:, 0])):4.2f} |Fy| {np.median(np.abs(o[:, 1])):4.2f} N")
for mode in ("wrench", "tracking"):
    meta = metas[mode]
    W = {k: v.astype(np.float64) for k, v in np.load(ROOT / f"tests/golden/{mode}_policy.npz").items()}
    mean, var = pins[f"{mode}_obs_rms_mean"], pins[f"{mode}_obs_rms_var"]
    lo, hi = pins[f"{mode}_action_low"], pins[f"{mode}_action_high"]
    def policy(obs, rng):
        x = np.clip((obs - mean) / np.sqrt(var + meta["epsilon"]), -meta["clip_obs"], meta["clip_obs"])
        h = np.tanh(x @ W["mlp_extractor.policy_net.0.weight"].T + W["mlp_extractor.policy_net.0.bias"])
        h = np.tanh(h @ W["mlp_extractor.policy_net.2.weight"].T + W["mlp_extractor.policy_net.2.bias"])
        mu = h @ W["action_net.weight"].T + W["action_net.bias"]
        return np.clip(mu + np.exp(W["log_std"]) * rng.standard_normal(mu.shape), lo, hi)
    print(f"{mode}: MuJoCo, end of training: {med(pins[mode + '_old_obs'])}")
    for scale in (1.0, 2.0, 4.0, 8.0):
        tc = 0.02 / np.sqrt(scale)
        o = Oracle(n, mode=MODE[mode], torso="top", seed=3, torso_solref_randomization=1, initial_probe_pos_randomization=1, early_termination=1, study_fix_tc=tc)
        rng = np.random.default_rng(0); obs = o.reset(); keep = []; rew = 0.0
        for k in range(T):
            obs, r, d, _, _ = o.step(policy(obs, rng)); rew += r.sum()
            if k >= T // 2 and k % 10 == 0:
                keep.append(obs.copy())
        print(f"  fix-row stiffness x {scale:3.0f} (time constant {tc * 1e3:4.1f} ms): {med(np.concatenate(keep))}; reward/step {rew / (n * T):.2f}")
