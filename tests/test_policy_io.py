"""CPU checks of the checkpoint readers in robotic-ultrasound-imaging_amd/policy.py against the committed fixtures."""
import importlib
import io
import json
import pickle
import zipfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_actor_critic_matches_sb3_layer_layout():
    pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")
    sd = {k: torch.from_numpy(v) for k, v in np.load(ROOT / "tests/golden/tracking_policy.npz").items()}
    net = pol.MlpActorCritic.from_sb3_state_dict(sd)
    x = torch.randn(5, 19)
    h = torch.tanh(x @ sd["mlp_extractor.policy_net.0.weight"].T + sd["mlp_extractor.policy_net.0.bias"])
    h = torch.tanh(h @ sd["mlp_extractor.policy_net.2.weight"].T + sd["mlp_extractor.policy_net.2.bias"])
    mean = h @ sd["action_net.weight"].T + sd["action_net.bias"]
    got, val = net(x)
    assert torch.allclose(got, mean, atol=1e-6) and val.shape == (5,)
    a = net.predict(x, deterministic=True, low=torch.zeros(6), high=torch.ones(6))
    assert a.min() >= 0 and a.max() <= 1
    assert torch.equal(net.log_std.data, sd["log_std"])


def test_sb3_zip_and_vecnormalize_readers_roundtrip(tmp_path):
    """Write a checkpoint in the SB3 on-disk format (zip{data json, policy.pth}; pickled VecNormalize) and read it back."""
    pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")
    sd = {k: torch.from_numpy(v) for k, v in np.load(ROOT / "tests/golden/tracking_policy.npz").items()}
    buf = io.BytesIO(); torch.save(sd, buf)
    zp = tmp_path / "model.zip"
    with zipfile.ZipFile(zp, "w") as z:
        z.writestr("data", json.dumps({"n_envs": 64, "gamma": 0.99}))
        z.writestr("policy.pth", buf.getvalue())
    sd2, data = pol.load_sb3_zip(zp)
    assert data["n_envs"] == 64 and all(torch.equal(sd[k], sd2[k]) for k in sd)
    # a VecNormalize pickle references stable_baselines3 classes; the reader provides stubs for them
    pol._stub("stable_baselines3.common.running_mean_std", "RunningMeanStd")
    pol._stub("stable_baselines3.common.vec_env.vec_normalize", "VecNormalize")
    import sys
    RMS = sys.modules["stable_baselines3.common.running_mean_std"].RunningMeanStd
    VN = sys.modules["stable_baselines3.common.vec_env.vec_normalize"].VecNormalize
    o, r, v = RMS(), RMS(), VN()
    o.__dict__.update(mean=np.arange(19.0), var=np.ones(19) * 2, count=10.0); r.__dict__.update(mean=3.0, var=4.0, count=10.0)
    v.__dict__.update(obs_rms=o, ret_rms=r, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8)
    pk = tmp_path / "vn.pkl"
    pk.write_bytes(pickle.dumps(v))
    st = pol.load_vecnormalize_pkl(pk)
    assert np.array_equal(st["obs_mean"], np.arange(19.0)) and st["ret_var"] == 4.0 and st["clip_obs"] == 10.0


def test_rollout_buffer_gae_matches_the_sb3_recursion():
    """DeviceRolloutBuffer against a literal numpy transcription of RolloutBuffer.compute_returns_and_advantage (SB3 1.1), and the
    flattening order of RolloutBuffer.swap_and_flatten"""
    pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")
    rng = np.random.default_rng(0)
    T, n, gamma, lam = 37, 5, 0.99, 0.95
    buf = pol.DeviceRolloutBuffer(T, n, obs_dim=19, act_dim=6, device="cpu", gamma=gamma, gae_lambda=lam)
    rew, val = rng.normal(size=(T, n)).astype(np.float32), rng.normal(size=(T, n)).astype(np.float32)
    starts = (rng.random((T, n)) < 0.1).astype(np.float32); starts[0] = 1
    obs = rng.normal(size=(T, n, 19)).astype(np.float32); act = rng.normal(size=(T, n, 6)).astype(np.float32)
    logp = rng.normal(size=(T, n)).astype(np.float32)
    for t in range(T):
        buf.add(*(torch.from_numpy(x[t]) for x in (obs, act, rew, starts, val, logp)))
    assert buf.full
    last_values, dones = rng.normal(size=n).astype(np.float32), (rng.random(n) < 0.3)
    buf.compute_returns_and_advantage(torch.from_numpy(last_values), torch.from_numpy(dones))
    adv = np.zeros((T, n), np.float64); last = np.zeros(n)
    for step in reversed(range(T)):
        if step == T - 1:
            nnt, nv = 1.0 - dones.astype(np.float64), last_values.astype(np.float64)
        else:
            nnt, nv = 1.0 - starts[step + 1], val[step + 1]
        delta = rew[step] + gamma * nv * nnt - val[step]
        last = delta + gamma * lam * nnt * last
        adv[step] = last
    assert np.allclose(buf.advantages.numpy(), adv, atol=2e-5) and np.allclose(buf.returns.numpy(), adv + val, atol=2e-5)
    got = list(buf.get(batch_size=None, generator=torch.Generator().manual_seed(1)))
    assert len(got) == 1 and got[0][0].shape == (T * n, 19)
    # flattened sample j = env j // T, step j % T (swap_and_flatten); the permutation is a bijection
    flat_obs = obs.transpose(1, 0, 2).reshape(T * n, 19)
    o = got[0][0].numpy()
    assert sorted(map(tuple, np.round(o, 5))) == sorted(map(tuple, np.round(flat_obs, 5)))
    sizes = [b[0].shape[0] for b in buf.get(batch_size=64)]
    assert sum(sizes) == T * n and max(sizes) == 64


def test_gaussian_policy_log_prob_and_entropy():
    pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")
    sd = {k: torch.from_numpy(v) for k, v in np.load(ROOT / "tests/golden/tracking_policy.npz").items()}
    net = pol.MlpActorCritic.from_sb3_state_dict(sd)
    x = torch.randn(7, 19)
    act, value, logp = net.sample(x, generator=torch.Generator().manual_seed(0))
    mean, v2 = net(x)
    dist = torch.distributions.Normal(mean, torch.exp(net.log_std).expand_as(mean))
    assert torch.allclose(logp, dist.log_prob(act).sum(-1), atol=1e-5) and torch.allclose(value, v2)
    v3, lp3, ent = net.evaluate_actions(x, act)
    assert torch.allclose(lp3, logp, atol=1e-5) and torch.allclose(ent, dist.entropy().sum(-1), atol=1e-5) and lp3.requires_grad
