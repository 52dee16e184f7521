"""P2PRolloutGather (device-to-device copies into IPC-mapped receive buffers instead of a collective kernel) with two processes.  The pool's
boxes have one GPU, so both ranks use cuda:0 -- RCCL cannot do that, the copy protocol can -- and the result is checked against the
single-process rollout of the whole batch, over more rounds than there are receive buffers."""
import importlib
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
N_LOCAL, T, WORLD, ROUNDS = 48, 8, 2, 5


def _worker(rank, port, out_dir):
    for p in (str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    usim = importlib.import_module("robotic-ultrasound-imaging_amd")
    d = importlib.import_module("robotic-ultrasound-imaging_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    kw = usim.default_robosuite_kwargs(); kw["horizon"] = 12
    env = usim.UltrasoundVecEnv(N_LOCAL, device=dev, seed=3, env_offset=rank * N_LOCAL, **kw)
    env.reset_tensor()
    g = d.P2PRolloutGather(device=dev)
    blocks = [env.alloc_block(T), env.alloc_block(T)]
    outs = []
    for r in range(ROUNDS):
        env.rollout_random(r * T, T, blocks[r & 1])
        if r:
            outs.append(g.wait().clone())                    # the previous round, gathered while this block was being simulated
        g.gather_async(blocks[r & 1])
    outs.append(g.wait().clone())
    torch.save(torch.stack(outs).cpu(), os.path.join(out_dir, f"rank{rank}.pt"))
    assert g.last_ms is None or g.last_ms >= 0
    g.close()
    dist.barrier()
    env.close()
    dist.destroy_process_group()


def test_p2p_gather_of_two_ranks_equals_the_whole_batch(usim, tmp_path):
    d = importlib.import_module("robotic-ultrasound-imaging_amd.distributed")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(WORLD)]           # [ROUNDS, world, T, n_local, C]
    assert got[0].shape == (ROUNDS, WORLD, T, N_LOCAL, 19 + 6 + 2) and torch.equal(got[0], got[1])
    kw = usim.default_robosuite_kwargs(); kw["horizon"] = 12
    env = usim.UltrasoundVecEnv(WORLD * N_LOCAL, device="cuda:0", seed=3, **kw)
    env.reset_tensor()
    blk = env.alloc_block(T)
    for r in range(ROUNDS):
        env.rollout_random(r * T, T, blk)
        ref = d.pack_block(blk).cpu()                                             # [T, world * n_local, C]
        full = torch.cat([got[0][r, k] for k in range(WORLD)], dim=1)
        assert torch.equal(full, ref), r
    assert d.unpack_block(ref, 6)["done"].any()
    env.close()
