#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched Ultrasound step at 4096 envs/GPU, random-action rollout
(BASELINE.json metric).  One process per GPU; for N > 1 the driver launches this file under torch.distributed.run -- and
`python bench.py --gpus N` without a launcher starts those N ranks itself (child processes, before anything touches a GPU).

A "step" is one env.step() of every environment of the rank (controller + forward dynamics + soft contact + sensors +
reward + termination + auto-reset), actions drawn in-kernel from the counter-based stream of BASELINE.md section 4,
transitions written into rollout blocks in HBM; with N > 1 each finished block is all-gathered over RCCL on a side
stream.  Prints ONE JSON line on rank 0."""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
# algorithmic bytes per env-step: compulsory fp32 state read+write with state resident in HBM
ALGO_BYTES = {"rigid": 316, "soft": 1912, "full": 4752}
WORKLOAD_NAME = {"rigid": "configs[1]: 4096 envs/GPU, rigid torso (contact solver off), OSC controller only",
                 "soft": "configs[2]: 4096 envs/GPU, soft-torso contact + force/velocity-tracking reward",
                 "randomised": "configs[4]: 8192 envs/GPU with domain-randomised torso stiffness/damping + probe friction",
                 "full": "configs[2] with the FULL torso (not the configuration the metric is quoted on): 270 elements on the free torso body, element-table contacts "
                         "(SURVEY.md 8 row a3 in full; csrc/usim_full.h, one wave per environment)"}


def cpu_baseline(workload, n_envs, budget_s=float(os.environ.get("USIM_CPU_BUDGET_S", "15"))):
    """The oracle (kind "port": the reference's own mujoco-py path cannot run) timed on this box's host
    cores with OpenMP over environments, on a bounded sample of the same workload."""
    import numpy as np
    from oracle_lib import Oracle, build_oracle
    native = True
    try:
        build_oracle(native=True)
    except Exception:
        native = False
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    ora = Oracle(n_envs, precision="f64", omp=True, native=native, torso={"soft": "top", "full": "full"}.get(workload, "none"), seed=3)
    ora.reset()
    acts = [ora.random_actions(k) for k in range(4)]
    t0 = time.perf_counter()
    for k in range(4):
        ora.step(acts[k])
    per_step = (time.perf_counter() - t0) / 4
    steps = int(max(8, min(2000, budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for k in range(steps):
        ora.step(acts[k % 4])
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of the same workload, fp64 C oracle, OpenMP over envs, "
                      f"{'-O3 -march=native' if native else '-O3'}"}


def _self_launch(gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as CHILD processes of a parent that has
    not touched a GPU (nothing above this line imports torch), exactly the way the driver would -- torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 -- and leave with the launcher's exit code.  (Never re-exec a process that has initialised the GPU.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL / device-tensor sharing across processes needs it on this image
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of this node (default: WORLD_SIZE of the launcher, else 1).  N > 1 without a launcher "
                    "around it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself")
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)        # (one refill period of the reset bank: the timed launches are then all full 256-step launches)
    ap.add_argument("--presteps", type=int, default=0, help="studies only: untimed steps between the synchronous reset and the warm-up (after one horizon the episodes of "
                    "the batch are de-synchronised; right after the reset every probe has just been pressed in and the first ~100 steps carry a third more contacts).  "
                    "Default 0: the run starts from the reset, as the W warm-up steps of the contract imply")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--workload", choices=["soft", "rigid", "full"], default="soft")
    ap.add_argument("--block", type=int, default=256, help="rollout block length T (steps per all-gather)")
    ap.add_argument("--randomize", action="store_true", help="BASELINE configs[4]: per-env randomised stiffness/damping + probe friction")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather", choices=["rccl", "p2p"], default="rccl", help="N > 1: RCCL all-gather (resident workgroups on the CUs) or peer-to-peer copies "
                    "of the packed block (copy engines, no CUs; distributed.P2PRolloutGather)")
    ap.add_argument("--pgs-iters", type=int, default=0, help="studies only: sweeps of the contact solver (default: the library's, usim_config.pgs_iters)")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="consecutive steps per kernel launch of the rollout (1 .. 256; default: the library's, 256)")
    ap.add_argument("--lanes-per-env", type=int, default=0, choices=[0, 1, 8, 16, 32, 64], help="kernel mapping: 16 lanes per environment (automatic), 8 (soft torso) or 1 (rigid torso)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launched and args.gpus > 1:
        sys.exit(_self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        # a 1-GPU number labelled n_gpus 1 must never answer a request for N
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / device-tensor sharing across processes on this image); set before torch loads HIP
    import torch
    import torch.distributed as dist
    # USIM_BENCH_STUB=<module in tests/>: the N > 1 plumbing of this file (self-launch, rendezvous, sharding by env_offset, the gather of every block,
    # max-over-ranks timing, the JSON line) on CPU tensors over gloo with a stand-in stepper -- tests/test_bench_launch.py; never a measurement
    stub = os.environ.get("USIM_BENCH_STUB")
    on_gpu = not stub
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an AMD GPU: the simulator has no CPU path")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    use_dist = world > 1 or os.environ.get("USIM_BENCH_FORCE_GATHER") == "1"      # the env var exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ranks_seen = dist.get_world_size() if use_dist else 1

    n = args.envs_per_gpu
    extra = {"friction_randomization": 1} if args.randomize else {}
    with_gather = use_dist and not args.no_gather
    # N > 1.  One default: the same kernels as for N = 1 on every rank and the RCCL all-gather of each finished rollout block on a side stream while
    # the next block is simulated.  One flag: --gather p2p moves the blocks with peer-to-peer copies into IPC-mapped buffers instead of a collective
    # kernel (distributed.P2PRolloutGather: no workgroups on the CUs next to the step kernels; validated with two processes on one GPU only).
    if args.lanes_per_env:
        extra["lanes_per_env"] = args.lanes_per_env
    if args.pgs_iters:
        extra["pgs_iters"] = args.pgs_iters
    if on_gpu:
        usim = importlib.import_module("robotic-ultrasound-imaging_amd")
        env = usim.UltrasoundVecEnv(n, device=device, seed=3, env_offset=rank * n, torso=args.workload, **extra, **usim.default_robosuite_kwargs())
    else:
        env = importlib.import_module(stub).StubEnv(n, env_offset=rank * n)
    dmod = importlib.import_module("robotic-ultrasound-imaging_amd.distributed")
    T = max(1, min(args.block, args.steps))
    blocks = [env.alloc_block(T), env.alloc_block(T)]      # double-buffered: gather block b while simulating b^1
    gather = (dmod.P2PRolloutGather(device=device) if args.gather == "p2p" else dmod.RolloutGather(device=device if on_gpu else None)) if with_gather else None
    if args.steps_per_launch:
        env.set_steps_per_launch(args.steps_per_launch)

    env.reset_tensor()
    step = 0
    done_p = 0
    while done_p < args.presteps:                          # state preparation (see --presteps): untimed, before the W warm-up steps
        k = min(T, args.presteps - done_p)
        env.rollout_random(step, k, blocks[0])
        step += k; done_p += k
    done_w = 0
    while done_w < args.warmup:                            # untimed warm-up steps
        k = min(T, args.warmup - done_w)
        env.rollout_random(step, k, blocks[0])
        step += k; done_w += k
    gathered = None
    if gather is not None:
        gather.gather_async(blocks[0]); gathered = gather.wait()

    def slice_block(blk, lo, hi):
        return {key: t[lo:hi] for key, t in blk.items()}

    def sync():
        if on_gpu:
            torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize(device)

    class _NoEvent:                                        # stub run: no device, no device events
        def record(self): pass
        def elapsed_time(self, other): return 0.0
    new_event = (lambda: torch.cuda.Event(enable_timing=True)) if on_gpu else _NoEvent

    # Everything the timed loop needs is built before the clock starts (events, the step-io blocks of every slice of both rollout blocks):
    # between the synchronise and the first launch the device is idle, and for a short run (the driver times 20 steps) every
    # microsecond of Python in there shows up in the result.
    n_blocks = (args.steps + T - 1) // T
    evs = [(new_event(), new_event()) for _ in range(n_blocks)]
    for e0, e1 in evs:                                     # torch creates the HIP event at its first record(): do that now, not in the timed region
        e0.record(); e1.record()
    io_cache = {}
    def io_of(bi, lo, hi):
        key = (bi, lo, hi)
        if key not in io_cache:
            io_cache[key] = env.block_io(slice_block(blocks[bi], lo, hi))
        return io_cache[key]
    for bi in (0, 1):
        io_of(bi, 0, T); io_of(bi, 0, args.steps - (n_blocks - 1) * T)
    rollout = env.rollout_random
    sync()
    refill_ms0, refill_n0 = env.refill_time()
    t0 = time.perf_counter()
    done_s, b, ib, n_gathers = 0, 0, 0, 0
    # kernel-duration leg of the roofline: HIP events around every block of step launches, recorded on the launch stream (the
    # simulator is launched on torch's current stream, so torch events are events of that stream); read after the timed region,
    # the host never blocks inside it
    while done_s < args.steps:                             # EXACTLY args.steps timed steps
        k = min(T, args.steps - done_s)
        e0, e1 = evs[ib]
        e0.record()
        rollout(step, k, io=io_of(b, 0, k))               # with the gather: simulate block b while block b^1 is in flight
        e1.record()
        if gather is not None:
            g = gather.wait()                              # block b^1 is gathered before the next iteration overwrites it
            gathered = gathered if g is None else g
            gather.gather_async(blocks[b])
            n_gathers += 1
        step += k; done_s += k; b ^= 1; ib += 1
    if gather is not None:
        gathered = gather.wait()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed                                # this rank's own clock (reported per rank for N > 1)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dev_ms = sum(a.elapsed_time(b_) for a, b_ in evs) if on_gpu else elapsed * 1e3
    # the event brackets contain the step launches and, every 256 steps, one reset-bank refill launch: take its device time out, so that
    # avg_kernel_us is the step kernel's (what a rocprofv3 kernel trace reports for it)
    refill_ms1, refill_n1 = env.refill_time()
    refill_ms, refill_n = refill_ms1 - refill_ms0, refill_n1 - refill_n0
    block_ms = dev_ms
    dev_ms = max(dev_ms - refill_ms, 0.0)
    if os.environ.get("USIM_BENCH_TRACE") == "1" and rank == 0:      # per-block device time, for diagnosis
        print("block us/step:", " ".join(f"{1e3 * a.elapsed_time(b_) / min(T, args.steps):.1f}" for a, b_ in evs), file=sys.stderr)
    kern_steps = args.steps
    avg_kernel_s = dev_ms * 1e-3 / kern_steps
    achieved_gbs = ALGO_BYTES[args.workload] * n / avg_kernel_s / 1e9

    traffic_step = None
    tfile = next((f for f in (ROOT / "profiles" / r / "traffic.json" for r in ("r06", "r05", "r04", "r03", "r02", "r01")) if f.exists()), ROOT / "profiles" / "r06" / "traffic.json")
    if tfile.exists() and n == 4096:
        try:
            tj = json.loads(tfile.read_text()).get(args.workload)
            if tj:
                # bytes per step of all environments, committed PMC profile of this command; FETCH_SIZE with the guide's gfx950 x2 correction
                traffic_step = (tj.get("fetch_kb_x2", 2 * tj["fetch_kb"]) + tj["write_kb"]) * 1024.0
        except Exception:
            traffic_step = None

    issue = None
    ifile = tfile.parent / "issue.json"
    if ifile.exists() and n == 4096:
        try:
            issue = json.loads(ifile.read_text()).get(args.workload)
        except Exception:
            issue = None

    # N > 1: what every rank measured, so that a first multi-GPU run can be read rank by rank (the headline value stays whole-job steps / max-over-ranks time)
    per_rank = None
    if use_dist:
        mine = torch.tensor([dev_ms * 1e3 / max(args.steps, 1), elapsed_local * 1e6 / max(args.steps, 1),
                             (gather.last_ms if (gather is not None and gather.last_ms is not None) else -1.0), float(block_ms * 1e3 / max(args.steps, 1))], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu().tolist()
        per_rank = {"avg_kernel_us": [r[0] for r in rows], "wall_us_per_step": [r[1] for r in rows], "gather_last_ms": [r[2] for r in rows], "block_us_per_step": [r[3] for r in rows]}

    # the mapping usim_create picks (csrc/usim_api.hip): soft torso -> the split kernel, 16-lane groups (32) up to 4096 envs, 8-lane groups (64) beyond
    lanes = int(extra.get("lanes_per_env", 0)) or ((32 if n <= 4096 else 64) if (args.workload == "soft" and not extra.get("waves_per_simd")) else 16)
    if args.workload == "full":
        lanes = 64                                                     # one wave per environment, one step per launch (csrc/usim_full.h)
    spl = env.steps_per_launch if (lanes in (16, 32, 64) and args.workload != "full") else 1        # consecutive steps per kernel launch (usim_set_steps_per_launch)
    spl = max(1, min(spl, T, args.steps))
    wl = "randomised" if (args.randomize and args.workload == "soft" and n == 8192) else args.workload
    if rank == 0:
        total_steps = args.steps * n * ranks_seen
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU, random-action rollout",
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_NAME[wl], "envs_per_gpu": n, "global_envs": n * ranks_seen,
                       "controller": "OSC_POSE impedance_mode=tracking", "rollout_block": T,
                       "presteps": args.presteps,           # untimed steps before the warm-up that de-synchronise the episodes after the synchronous reset (0: none)
                       "domain_randomisation": "stiffness+damping" + ("+friction" if args.randomize else ""),
                       "parallelism": f"env-shard x{world}" + ("" if gather is None else (" + RCCL all-gather of transition blocks" if args.gather == "rccl" else
                                                                 " + peer-to-peer copies of transition blocks (copy engines)")),
                       "steps_per_launch": spl, "lanes_per_env": lanes, "waves_per_simd": int(extra.get("waves_per_simd", 0)) or "auto",
                       "contact_solver": ("block Gauss-Seidel (continuous local solve per visit) over the probe and the element-table contacts, explicit pair of coincident probe contacts" if args.workload == "full" else
                                          "block Jacobi + line search (slope taken block by block), explicit pair of coincident probe contacts (usim_config.pair_model 1)"),
                       "contact_solver_iterations": int(extra.get("pgs_iters", 0)) or "default (24)"},
            # what actually ran: the ranks the process group saw (never the --gpus argument), and the exchange step of the N > 1 path
            "ranks_seen": ranks_seen,
            "gather": None if gather is None else {"kind": args.gather, "backend": dist.get_backend(), "blocks_in_timed_region": n_gathers,
                                                   "last_ms": gather.last_ms, "result_shape": None if gathered is None else list(gathered.shape),
                                                   # the gather of block b runs on a side stream while block b + 1 is simulated: it is hidden when it is shorter than a block
                                                   "block_ms": block_ms / max(n_blocks, 1), "hidden_behind_next_block": None if gather.last_ms is None else bool(gather.last_ms < block_ms / max(n_blocks, 1))},
            # one entry per rank (N > 1): step-kernel time, this rank's own wall clock, its last gather, its block time
            "per_rank": per_rank,
            "per_rank_spread": None if per_rank is None else {k: [min(v), max(v)] for k, v in per_rank.items()},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         # HBM bytes per launch from the committed PMC profile (FETCH_SIZE with the guide's gfx950 x2 correction + WRITE_SIZE).  Far below the
                         # algorithmic bytes: inside a multi-step launch the state of the 4096 environments (7.8 MB) stays in the XCDs' L2 from one step
                         # to the next, and what reaches the memory side is mostly the transition block
                         "traffic": None if traffic_step is None else traffic_step * spl, "traffic_per_step": traffic_step,
                         "traffic_source": f"{tfile.relative_to(ROOT)} (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per step of all environments, times steps_per_launch)" if traffic_step else None,
                         # one launch advances all n environments by `steps_per_launch` steps; achieved = algorithmic bytes per launch / launch duration
                         "algorithmic_bytes_per_launch": ALGO_BYTES[args.workload] * n * spl, "algorithmic_bytes_per_env_step": ALGO_BYTES[args.workload],
                         "steps_per_launch": spl, "avg_launch_us": avg_kernel_s * 1e6 * spl,
                         "refill_launches": refill_n, "refill_us_per_step": refill_ms * 1e3 / args.steps, "block_us_per_step": block_ms * 1e3 / args.steps,
                         "avg_kernel_us": avg_kernel_s * 1e6, "kernel": ("usim_step_kernel<2, 64, 0> (full torso: one wave per environment)" if args.workload == "full" else
                                    {32: "usim_step32_kernel", 64: "usim_step32_kernel (8-lane groups)", 16: "usim_step16_kernel"}.get(lanes, "usim_step_kernel") + ("<multi-step>" if spl > 1 else "")),
                         # what binds the kernel, from the committed PMC profile of this command (tools/profile.sh -> profiles/<round>/issue.json): vector instructions
                         # issued per wave and step, and the share of the waves' cycles in which one issues -- the kernel is instruction-issue bound, not HBM bound
                         "issue": issue,
                         # (`traffic`, `traffic_per_step` and `issue` are QUOTED from the committed profile named here -- its own command line, not this run's)
                         "issue_source_config": None if issue is None else {"file": str(ifile.relative_to(ROOT)), "command": "bench.py --steps 2048 --warmup 256 (tools/profile.sh)", "steps_per_launch": 256},
                         "note": ("latency-bound, not HBM-bound: the step of an environment is a chain of (contacts x sweeps) Gauss-Seidel visits -- ~70 contacts (8 probe pairs, ~54 "
                                  "element-table contacts of the resting torso) x 24 sweeps x ~0.7 us, warm-started from the previous step -- run by one wave; 19.4 KB of LDS and 256 registers per environment: eight of them on a CU, two waves per SIMD; "
                                  "see DESIGN.md section 4.11" if args.workload == "full" else
                                  "latency-bound, not HBM-bound: one environment is a serial instruction chain; a wave issues one instruction per ~4 cycles whatever its lanes do while a "
                                  "SIMD would take four such waves (profiles/r05/micro_two_wave.txt), and at 4096 envs the 512 registers per SIMD lane hold two; the state of a multi-step "
                                  "launch lives in registers and L2; see DESIGN.md section 5")},
        }
        if world == 1 and on_gpu and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, n)
        result = json.dumps(out)
    else:
        result = None
    if gather is not None:
        gather.close()
    env.close()
    if use_dist:
        dist.destroy_process_group()
    if result is not None:
        # RCCL writes its version banner through C stdio; flush that first so that the JSON record is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(result, flush=True)


if __name__ == "__main__":
    main()
