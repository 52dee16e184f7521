"""Replay of the reference's trained checkpoints on the CPU oracle (tests/golden/*_policy.npz, VecNormalize statistics from reference_pins.npz),
evaluated as src/rl.py:171-192 does: reward per step, episode length, and the medians of the in-contact samples that the end-of-training `old_obs`
of the reference's VecNormalize pickles give for MuJoCo.   usage: python tests/studies/replay_oracle.py [n] [steps] [modes] [key=value ...]"""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from oracle_lib import Oracle

MODE = {"tracking": 0, "variable_z": 2, "wrench": 3}
REF = {"tracking": (8.12, 727), "variable_z": (8.03, 718), "wrench": (8.61, 440)}


def med(o):
    o = o[o[:, 2] > 0]
    return (f"Fz {np.median(o[:, 2]):5.2f} N, height {np.median(o[:, 14]) * 1e3:5.2f} mm ({np.quantile(o[:, 14], .25) * 1e3:5.2f} .. {np.quantile(o[:, 14], .75) * 1e3:5.2f}), "
            f"|Fx| {np.median(np.abs(o[:, 0])):4.2f} |Fy| {np.median(np.abs(o[:, 1])):4.2f} N")


def replay(mode, n, T, deterministic=False, **kw):
    kw = dict(kw)
    metas = json.loads((ROOT / "tests/golden/reference_pins.json").read_text())
    pins = np.load(ROOT / "tests/golden/reference_pins.npz")
    meta = metas[mode]
    W = {k: v.astype(np.float64) for k, v in np.load(ROOT / f"tests/golden/{mode}_policy.npz").items()}
    mean, var = pins[f"{mode}_obs_rms_mean"], pins[f"{mode}_obs_rms_var"]
    lo, hi = pins[f"{mode}_action_low"], pins[f"{mode}_action_high"]

    def policy(obs, rng):
        x = np.clip((obs - mean) / np.sqrt(var + meta["epsilon"]), -meta["clip_obs"], meta["clip_obs"])
        h = np.tanh(x @ W["mlp_extractor.policy_net.0.weight"].T + W["mlp_extractor.policy_net.0.bias"])
        h = np.tanh(h @ W["mlp_extractor.policy_net.2.weight"].T + W["mlp_extractor.policy_net.2.bias"])
        mu = h @ W["action_net.weight"].T + W["action_net.bias"]
        return np.clip(mu if deterministic else mu + np.exp(W["log_std"]) * rng.standard_normal(mu.shape), lo, hi)

    o = Oracle(n, omp=True, mode=MODE[mode], torso=kw.pop("torso", "top"), seed=3, torso_solref_randomization=1, initial_probe_pos_randomization=1, early_termination=1, **kw)
    rng = np.random.default_rng(0); obs = o.reset(); keep = []; rew = 0.0
    ep_len = np.zeros(n); lens = []; causes = np.zeros(32, int)
    for k in range(T):
        obs, r, d, _, _ = o.step(policy(obs, rng)); rew += r.sum(); ep_len += 1
        if d.any():
            lens += list(ep_len[d]); ep_len[d] = 0
            causes += np.bincount(o.last_info()["cause"][d], minlength=32)
        if k >= T // 4 and k % 10 == 0:
            keep.append(obs.copy())
    return {"reward_per_step": rew / (n * T), "ep_len": float(np.mean(lens)) if lens else float("nan"), "episodes": len(lens), "med": med(np.concatenate(keep)),
            "old": med(pins[mode + "_old_obs"]), "causes": {c: int(v) for c, v in enumerate(causes) if v}}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else list(MODE)
    kw = {}
    for a in sys.argv[4:]:
        k, v = a.split("=")
        kw[k] = v if k == "torso" else (int(v) if v.lstrip("-").isdigit() else float(v))
    for m in modes:
        t0 = time.time()
        r = replay(m, n, T, **kw)
        print(f"{m:10s} {kw}: reward/step {r['reward_per_step']:.2f} (MuJoCo {REF[m][0]}), episode length {r['ep_len']:.0f} (MuJoCo {REF[m][1]}; {r['episodes']} episodes, causes {r['causes']})"
              f"\n   here   {r['med']}\n   MuJoCo {r['old']}   [{time.time() - t0:.0f} s]", flush=True)
