import ctypes, sys, time, importlib
mode = sys.argv[1]
if mode == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(1))
sys.path.insert(0, "/root/repo")
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
env = usim.UltrasoundVecEnv(4096, torso="soft", **usim.default_robosuite_kwargs())
blk = env.alloc_block(20); io = env.block_io(blk)
env.reset_tensor(); env.rollout_random(0, 5, io=io); torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); env.rollout_random(5 + 20 * rep, 20, io=io); e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(mode, f"wall {(t1 - t0) * 1e6:.0f} us, events {e0.elapsed_time(e1) * 1e3:.0f} us, overhead {(t1 - t0) * 1e6 - e0.elapsed_time(e1) * 1e3:.0f} us")
