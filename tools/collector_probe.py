"""Policy in the loop: env-steps/s of policy.collect_rollouts (eager, ~40 launches per step from Python) against policy.GraphedCollector (the same
rollout recorded once as a HIP graph), and that the two produce the same buffer.   usage: python tools/collector_probe.py [n_envs] [T] [repeats]"""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")

def make():
    torch.manual_seed(0)
    env = usim.UltrasoundVecEnv(n, device="cuda:0", seed=3, **usim.default_robosuite_kwargs())
    policy = pol.MlpActorCritic(19, env.action_dim).to(dev)
    vn = pol.DeviceVecNormalize(n, 19, device=dev, training=True, norm_reward=True)
    buf = pol.DeviceRolloutBuffer(T, n, 19, env.action_dim, device=dev)
    return env, policy, vn, buf

import os
FUSED_ONLY = bool(os.environ.get('FUSED_ONLY'))
# --- same numbers?  (default generator re-seeded identically; the graphed collector runs `warmup_steps` eager steps first, so does the eager side)
if not FUSED_ONLY:
    env, policy, vn, buf = make()
    torch.manual_seed(1)
    gc = pol.GraphedCollector(env, policy, vn, buf, warmup_steps=2)
    gc.collect(); torch.cuda.synchronize()
    g_obs, g_adv, g_mean = buf.observations.clone(), buf.advantages.clone(), vn.obs_mean.clone()
    env2, policy2, vn2, buf2 = make()
    policy2.load_state_dict(policy.state_dict())
    torch.manual_seed(1)
    obs, start = env2.reset_tensor().clone(), torch.ones(n, dtype=torch.bool, device=dev)
    low, high = torch.as_tensor(env2.action_space.low, device=dev), torch.as_tensor(env2.action_space.high, device=dev)
    with torch.no_grad():
        for _ in range(2):                                                   # the warm-up steps
            nobs = vn2.normalize_obs(obs); act, value, logp = policy2.sample(nobs)
            o, rew, done = env2.step_tensor(torch.max(torch.min(act, high), low)); vn2.normalize_reward(rew, done)
            obs, start = o.clone(), done.bool().clone()
    env2.refill_bank()
    pol.collect_rollouts(env2, policy2, vn2, buf2, obs=obs, episode_start=start); torch.cuda.synchronize()
    print("graph == eager: observations", bool(torch.equal(g_obs, buf2.observations)), " advantages", bool(torch.equal(g_adv, buf2.advantages)),
          " obs statistics", bool(torch.equal(g_mean, vn2.obs_mean)), " max |d adv|", float((g_adv - buf2.advantages).abs().max()))

    # --- speed
    for label, fn in (("eager  collect_rollouts", lambda: pol.collect_rollouts(env2, policy2, vn2, buf2, obs=obs, episode_start=start)), ("graphed collector     ", gc.collect)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{label}: {dt * 1e3:8.2f} ms per rollout of {n} x {T}  = {n * T / dt / 1e6:7.2f} M env-steps/s  ({dt / T * 1e6:6.1f} us per step)")
# the fused kernels (usim_policy_step / usim_policy_reward / usim_policy_gae), recorded as a graph
env3, policy3, vn3, buf3 = make()
fr = pol.FusedRollout(env3, policy3, vn3, buf3, seed=1, fused_stats=n <= 8192)      # (two launches per step up to the library's limit)
fr.collect(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fr.collect()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"fused kernels + graph  : {dt * 1e3:8.2f} ms per rollout of {n} x {T}  = {n * T / dt / 1e6:7.2f} M env-steps/s  ({dt / T * 1e6:6.1f} us per step)")
f3 = pol.FusedRollout(env3, policy3, vn3, buf3, seed=1, fused_stats=False)
f3.collect(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    f3.collect()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"  (three launches/step): {dt * 1e3:8.2f} ms per rollout of {n} x {T}  = {n * T / dt / 1e6:7.2f} M env-steps/s  ({dt / T * 1e6:6.1f} us per step)")
fe = pol.FusedRollout(env3, policy3, vn3, buf3, seed=1, graph=False, fused_stats=n <= 8192)
fe.collect(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fe.collect()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"fused kernels, eager   : {dt * 1e3:8.2f} ms per rollout of {n} x {T}  = {n * T / dt / 1e6:7.2f} M env-steps/s  ({dt / T * 1e6:6.1f} us per step)")
# the simulator alone, for scale
env = env3
blk = env.alloc_block(T)
env.rollout_random(0, T, blk); torch.cuda.synchronize()
t0 = time.perf_counter(); env.rollout_random(T, T, blk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"simulator alone (random actions, multi-step launches): {n * T / dt / 1e6:7.2f} M env-steps/s")
t0 = time.perf_counter()
for k in range(T):
    env.step_tensor(env.random_actions_tensor(k))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"simulator alone (usim_step per step from Python):      {n * T / dt / 1e6:7.2f} M env-steps/s")
