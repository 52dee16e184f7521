"""Learnability check: a from-scratch PPO (clip objective, GAE, Adam; SB3's default hyper-parameters where they matter) trained on the
batched simulator through the device-side collector of policy.py.  Not part of the product path (the reference's training loop is
stable-baselines3 and stays the caller); it answers one question: does reward per step climb from the random-gain level (5.6) towards
what the reference's own policy earns on it (8.1)?   usage: python tools/ppo_demo.py [iterations] [n_envs] [n_steps] [impedance_mode] [collector]
collector: eager (policy.collect_rollouts, ~100 PyTorch launches per step), graph (the same loop recorded as a HIP graph) or fused (the library's policy
kernels, usim_policy_step / _reward / _gae, recorded as a HIP graph; the default)"""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
pol = importlib.import_module("robotic-ultrasound-imaging_amd.policy")

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
mode = sys.argv[4] if len(sys.argv) > 4 else "tracking"
collector = sys.argv[5] if len(sys.argv) > 5 else "fused"
torch.manual_seed(0)
kw = usim.default_robosuite_kwargs(); kw["controller_configs"] = dict(kw["controller_configs"], impedance_mode=mode)
env = usim.UltrasoundVecEnv(n, device="cuda:0", seed=3, **kw)
print(f"mode {mode}, {n} envs x {T} steps per iteration", flush=True)
dev = env.device
policy = pol.MlpActorCritic(19, env.action_dim).to(dev)
for m in policy.modules():                                      # SB3: orthogonal init, gain sqrt(2) (0.01 for the action head, 1 for the value head)
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.orthogonal_(m.weight, gain=2 ** 0.5); torch.nn.init.zeros_(m.bias)
torch.nn.init.orthogonal_(policy.action_net.weight, gain=0.01); torch.nn.init.orthogonal_(policy.value_net.weight, gain=1.0)
vn = pol.DeviceVecNormalize(n, 19, device=dev, training=True, norm_reward=True)
buf = pol.DeviceRolloutBuffer(T, n, 19, env.action_dim, device=dev)
opt = torch.optim.Adam(policy.parameters(), lr=3e-4, eps=1e-5)
gen = torch.Generator(device=dev); gen.manual_seed(0)
obs, start = None, None
coll = pol.FusedRollout(env, policy, vn, buf, seed=0) if collector == "fused" else (pol.GraphedCollector(env, policy, vn, buf) if collector == "graph" else None)
print(f"collector: {collector}", flush=True)
torch.cuda.synchronize()
t0 = time.time()
t_collect = 0.0
for it in range(iters):
    torch.cuda.synchronize(); tc = time.time()
    if coll is not None:
        coll.collect()
        raw = coll.raw_reward_sum
    else:
        # raw reward per step of this iteration: read from the env while collecting (the buffer holds normalised rewards)
        raw = torch.zeros((), dtype=torch.float64, device=dev)
        step_tensor = env.step_tensor
        def counting_step(a, _f=step_tensor):
            o, r, d = _f(a); raw.add_(r.sum()); return o, r, d
        env.step_tensor = counting_step
        obs, start = pol.collect_rollouts(env, policy, vn, buf, obs=obs, episode_start=start, generator=gen)
        env.step_tensor = step_tensor
    torch.cuda.synchronize(); t_collect += time.time() - tc
    adv_all = buf.advantages
    for epoch in range(4):
        for ob, ac, val, lp, adv, ret in buf.get(batch_size=n * T // 8, generator=gen):
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            values, logp, ent = policy.evaluate_actions(ob, ac)
            ratio = torch.exp(logp - lp)
            pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 0.8, 1.2)).mean()
            vf = torch.nn.functional.mse_loss(values, ret)
            loss = pg + 0.5 * vf - 0.0 * ent.mean()
            opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(policy.parameters(), 0.5); opt.step()
    print(f"iter {it:3d}  env-steps {(it + 1) * n * T:9d}  reward/step {float(raw) / (n * T):6.3f}  value loss {float(vf.detach()):7.4f}  "
          f"std {float(torch.exp(policy.log_std.detach()).mean()):5.3f}  wall {time.time() - t0:6.1f}s  (collecting {t_collect:5.2f}s)", flush=True)
print(f"collection: {iters * n * T / t_collect / 1e6:.1f} M env-steps/s; with the PPO updates: {iters * n * T / (time.time() - t0) / 1e6:.2f} M env-steps/s")
env.close()
