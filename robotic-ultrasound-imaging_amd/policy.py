"""Replay of the reference's trained stable-baselines3 PPO policies on the batched simulator, entirely on the GPU
(SURVEY.md section 8f rank 1).

The reference evaluates a checkpoint with `VecNormalize.load(...)`, `PPO.load(...)`, `model.predict(obs)` (src/rl.py:171-192).
The shipped artefacts are `trained_rl_models/<name>.zip` (SB3 1.1.0a5: json `data` + `policy.pth` state dict of an
ActorCriticPolicy with net_arch [dict(pi=[256,128], vf=[256,128])], tanh activations) and `vec_normalize_<name>.pkl`
(obs_rms / ret_rms, clip 10, gamma 0.99, eps 1e-8).  Neither SB3 nor gym is imported: the zip is read with zipfile +
torch.load(weights_only=True), the pickle with stub classes."""
import io
import json
import pickle
import sys
import types
import zipfile

import numpy as np
import torch


def _stub(modname, clsname):
    parts = modname.split(".")
    for i in range(1, len(parts) + 1):
        name = ".".join(parts[:i])
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules[modname], clsname):
        setattr(sys.modules[modname], clsname,
                type(clsname, (), {"__module__": modname, "__setstate__": lambda self, st: self.__dict__.update(st)}))


def load_sb3_zip(path):
    """-> (state_dict of CPU tensors, data dict) of a stable-baselines3 PPO checkpoint"""
    with zipfile.ZipFile(path) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        data = json.loads(z.read("data"))
    return sd, data


def load_vecnormalize_pkl(path):
    """-> dict(obs_mean, obs_var, count, ret_mean, ret_var, clip_obs, clip_reward, gamma, epsilon)"""
    for mod, cls in (("gym.spaces.box", "Box"), ("gym.spaces.space", "Space"),
                     ("stable_baselines3.common.running_mean_std", "RunningMeanStd"),
                     ("stable_baselines3.common.vec_env.vec_normalize", "VecNormalize")):
        _stub(mod, cls)
    with open(path, "rb") as f:
        v = pickle.load(f).__dict__
    o, r = v["obs_rms"].__dict__, v["ret_rms"].__dict__
    return {"obs_mean": np.asarray(o["mean"], dtype=np.float64), "obs_var": np.asarray(o["var"], dtype=np.float64), "count": float(o["count"]),
            "ret_mean": float(r["mean"]), "ret_var": float(r["var"]), "clip_obs": float(v["clip_obs"]), "clip_reward": float(v["clip_reward"]),
            "gamma": float(v["gamma"]), "epsilon": float(v["epsilon"])}


class DeviceVecNormalize:
    """stable_baselines3 VecNormalize on device tensors: running mean/variance of observations and discounted returns
    (parallel-variance update of RunningMeanStd), normalisation with clipping.  `training=False` freezes the statistics
    (src/rl.py:180-181)."""

    def __init__(self, num_envs, obs_dim=19, device="cuda:0", clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8, training=True, norm_reward=True):
        dev = torch.device(device)
        self.obs_mean = torch.zeros(obs_dim, dtype=torch.float64, device=dev)
        self.obs_var = torch.ones(obs_dim, dtype=torch.float64, device=dev)
        self.obs_count = 1e-4
        self.ret_mean = torch.zeros((), dtype=torch.float64, device=dev)
        self.ret_var = torch.ones((), dtype=torch.float64, device=dev)
        self.ret_count = 1e-4
        self.returns = torch.zeros(num_envs, dtype=torch.float64, device=dev)
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.training, self.norm_reward = training, norm_reward

    @classmethod
    def from_stats(cls, stats, num_envs, device="cuda:0", training=False, norm_reward=False):
        self = cls(num_envs, len(stats["obs_mean"]), device, stats["clip_obs"], stats["clip_reward"], stats["gamma"], stats["epsilon"], training, norm_reward)
        self.obs_mean = torch.as_tensor(stats["obs_mean"], dtype=torch.float64, device=self.obs_mean.device)
        self.obs_var = torch.as_tensor(stats["obs_var"], dtype=torch.float64, device=self.obs_mean.device)
        self.obs_count = stats["count"]
        self.ret_mean = torch.as_tensor(stats["ret_mean"], dtype=torch.float64, device=self.obs_mean.device)
        self.ret_var = torch.as_tensor(stats["ret_var"], dtype=torch.float64, device=self.obs_mean.device)
        self.ret_count = stats["count"]
        return self

    @staticmethod
    def _update(mean, var, count, batch):
        b = batch.to(torch.float64)
        bm, bv, bn = b.mean(0), b.var(0, unbiased=False), b.shape[0]
        delta, tot = bm - mean, count + bn
        new_mean = mean + delta * bn / tot
        m2 = var * count + bv * bn + delta * delta * count * bn / tot
        return new_mean, m2 / tot, tot

    def normalize_obs(self, obs):
        if self.training:
            self.obs_mean, self.obs_var, self.obs_count = self._update(self.obs_mean, self.obs_var, self.obs_count, obs)
        out = (obs.to(torch.float64) - self.obs_mean) / torch.sqrt(self.obs_var + self.epsilon)
        return torch.clamp(out, -self.clip_obs, self.clip_obs).to(torch.float32)

    def normalize_reward(self, rew, done):
        if self.training:
            self.returns = self.returns * self.gamma + rew.to(torch.float64)
            self.ret_mean, self.ret_var, self.ret_count = self._update(self.ret_mean, self.ret_var, self.ret_count, self.returns)
            self.returns = torch.where(done.bool(), torch.zeros_like(self.returns), self.returns)
        if not self.norm_reward:
            return rew
        return torch.clamp(rew.to(torch.float64) / torch.sqrt(self.ret_var + self.epsilon), -self.clip_reward, self.clip_reward).to(torch.float32)


class MlpActorCritic(torch.nn.Module):
    """SB3 ActorCriticPolicy(MlpPolicy, net_arch=[dict(pi=[256,128], vf=[256,128])]) with tanh activations and a state-independent
    log_std (rl_config.yaml:11-15; layer names as in the checkpoints' policy.pth)."""

    def __init__(self, obs_dim=19, act_dim=6, pi=(256, 128), vf=(256, 128)):
        super().__init__()
        def mlp(sizes):
            layers, d = [], obs_dim
            for h in sizes:
                layers += [torch.nn.Linear(d, h), torch.nn.Tanh()]
                d = h
            return torch.nn.Sequential(*layers)
        self.policy_net, self.value_net_body = mlp(pi), mlp(vf)
        self.action_net = torch.nn.Linear(pi[-1], act_dim)
        self.value_net = torch.nn.Linear(vf[-1], 1)
        self.log_std = torch.nn.Parameter(torch.zeros(act_dim))

    @classmethod
    def from_sb3_state_dict(cls, sd):
        act_dim, obs_dim = sd["action_net.weight"].shape[0], sd["mlp_extractor.policy_net.0.weight"].shape[1]
        self = cls(obs_dim, act_dim, (sd["mlp_extractor.policy_net.0.weight"].shape[0], sd["mlp_extractor.policy_net.2.weight"].shape[0]),
                   (sd["mlp_extractor.value_net.0.weight"].shape[0], sd["mlp_extractor.value_net.2.weight"].shape[0]))
        mapped = {}
        for k, v in sd.items():
            k2 = k.replace("mlp_extractor.policy_net.", "policy_net.").replace("mlp_extractor.value_net.", "value_net_body.")
            mapped[k2] = torch.as_tensor(v)
        self.load_state_dict(mapped)
        return self

    def forward(self, obs):
        return self.action_net(self.policy_net(obs)), self.value_net(self.value_net_body(obs)).squeeze(-1)

    @torch.no_grad()
    def predict(self, obs, deterministic=True, low=None, high=None, generator=None):
        """PPO.predict: Gaussian mean (or a sample), clipped to the action box like SB3 does before env.step"""
        mean, _ = self.forward(obs)
        act = mean if deterministic else mean + torch.exp(self.log_std) * torch.randn(mean.shape, device=mean.device, generator=generator)
        if low is not None:
            act = torch.max(torch.min(act, high), low)
        return act


@torch.no_grad()
def policy_rollout(env, policy, vecnorm, steps, deterministic=False, seed=0):
    """Run `steps` env steps of the policy on the device-resident fast path (no host synchronisation inside the loop).
    Returns raw-observation running statistics (mean, var over all visited steps), mean reward per step, and episode stats."""
    dev = env.device
    low = torch.as_tensor(env.action_space.low, device=dev)
    high = torch.as_tensor(env.action_space.high, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    obs = env.reset_tensor()
    n = env.num_envs
    s1 = torch.zeros(obs.shape[1], dtype=torch.float64, device=dev); s2 = torch.zeros_like(s1)
    rew_sum = torch.zeros((), dtype=torch.float64, device=dev)
    ep_n = torch.zeros((), dtype=torch.float64, device=dev); ep_r = torch.zeros_like(ep_n); ep_l = torch.zeros_like(ep_n)
    for _ in range(steps):
        o = obs.to(torch.float64)
        s1 += o.sum(0); s2 += (o * o).sum(0)
        act = policy.predict(vecnorm.normalize_obs(obs), deterministic, low, high, gen)
        obs, rew, done = env.step_tensor(act)
        vecnorm.normalize_reward(rew, done)
        rew_sum += rew.sum()
        d = done.bool()
        ep_n += d.sum(); ep_r += env.episode_return[d].sum(); ep_l += env.episode_length[d].sum()
    tot = steps * n
    mean = s1 / tot
    out = {"obs_mean": mean.cpu().numpy(), "obs_var": (s2 / tot - mean * mean).cpu().numpy(), "reward_per_step": float(rew_sum / tot),
           "episodes": float(ep_n), "mean_episode_return": float(ep_r / ep_n) if ep_n > 0 else float("nan"),
           "mean_episode_length": float(ep_l / ep_n) if ep_n > 0 else float("nan")}
    return out
