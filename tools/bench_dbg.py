import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
variant = sys.argv[1]
device = torch.device("cuda", 0); torch.cuda.set_device(0)
env = usim.UltrasoundVecEnv(4096, device=device, seed=3, env_offset=0, torso="soft", **usim.default_robosuite_kwargs())
T = 20
blocks = [env.alloc_block(T), env.alloc_block(T)]
env.reset_tensor()
env.rollout_random(0, 5, blocks[0])
torch.cuda.synchronize(device)
if variant == "sleep": time.sleep(0.05)
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
if variant == "slice":
    blk = {k: t[0:20] for k, t in blocks[0].items()}
else:
    blk = blocks[0]
env.rollout_random(5, 20, blk)
e1.record()
th = time.perf_counter()
torch.cuda.synchronize(device)
t1 = time.perf_counter()
print(variant, f"events {1e3 * e0.elapsed_time(e1) / 20:.1f} us/step, wall {(t1 - t0) * 1e6 / 20:.1f} us/step, host enqueue {(th - t0) * 1e6:.0f} us")
