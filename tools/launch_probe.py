"""Host enqueue time vs device time of short bursts of step launches (the driver's bench run is 5 + 20 steps)."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
for torso in ("soft", "rigid"):
    env = usim.UltrasoundVecEnv(4096, torso=torso, **usim.default_robosuite_kwargs())
    env.reset_tensor(); blk = env.alloc_block(128)
    env.rollout_random(0, 5, blk); torch.cuda.synchronize()
    for burst in (20, 20, 20, 128, 128):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(); env.rollout_random(1000, burst, blk); e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{torso} burst {burst:4d}: host enqueue {1e6 * (t1 - t0) / burst:6.2f} us/launch, wall {1e6 * (t2 - t0) / burst:6.2f} us/step, device (events) {1e3 * e0.elapsed_time(e1) / burst:6.2f} us/step", flush=True)
    env.close()

# is the slow start after a synchronize a clock ramp?  20 steps timed (a) right behind 500 untimed steps without a host sync in between,
# (b) after a synchronize that follows those 500 steps immediately, (c) after a synchronize and 5 ms of host sleep
env = usim.UltrasoundVecEnv(4096, torso="soft", **usim.default_robosuite_kwargs())
env.reset_tensor(); blk = env.alloc_block(128)
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout_random(5000, n, blk); e1.record()
    return e0, e1
for label, prep in (("hot, no sync", lambda: None), ("sync", lambda: torch.cuda.synchronize()), ("sync + 5 ms idle", lambda: (torch.cuda.synchronize(), time.sleep(0.005))),
                    ("sync + 100 ms idle", lambda: (torch.cuda.synchronize(), time.sleep(0.1)))):
    for rep in range(2):
        for _ in range(4): env.rollout_random(2000, 128, blk)
        prep()
        e0, e1 = timed(20)
        torch.cuda.synchronize()
        print(f"{label:20s}: {1e3 * e0.elapsed_time(e1) / 20:6.2f} us/step", flush=True)
