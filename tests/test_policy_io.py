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
