"""Multi-GPU sharding of the batched simulator: one process per GPU, environments partitioned by global index, no
communication inside step(); one RCCL all-gather of each rollout block.

The reference's only "collective" is SubprocVecEnv's gather of (obs, reward, done) tuples from 64 worker processes
over pipes every step (src/rl.py:130).  Here the unit is a block of T transitions per GPU, gathered with
torch.distributed (backend "nccl" is RCCL on ROCm; on MI355X the 8 GPUs are fully connected by xGMI so the
all-gather is 7 concurrent point-to-point transfers per GPU).  The gather runs on a side stream so that it overlaps
the simulation of the next block.  The helpers are backend-agnostic (the CPU tests drive them with gloo)."""
import torch
import torch.distributed as dist


def shard_range(total_envs, world_size, rank):
    """Global env range [lo, hi) simulated by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_block(block):
    """[T, n, 19 + A + 2] float32: obs | act | rew | done -- one tensor per block keeps the all-gather a single
    large message (bucket size matters on per-link-bound xGMI)."""
    parts = [block["obs"]]
    if "act" in block:
        parts.append(block["act"])
    parts += [block["rew"].unsqueeze(-1), block["done"].to(torch.float32).unsqueeze(-1)]
    return torch.cat(parts, dim=-1).contiguous()


def unpack_block(packed, action_dim):
    o = 19
    out = {"obs": packed[..., :o]}
    if action_dim:
        out["act"] = packed[..., o:o + action_dim]
    out["rew"] = packed[..., o + action_dim]
    out["done"] = packed[..., o + action_dim + 1] > 0.5
    return out


class RolloutGather:
    """All-gather of rollout blocks across the ranks of `group`.

    gather(block) returns a [world, T, n_local, C] tensor (every rank holds the full batch, as PPO's update
    needs).  With a CUDA/HIP device the collective is issued on a dedicated side stream: call gather_async() right
    after a block has been enqueued, keep simulating the next block, and wait() before consuming the result."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device
        self._stream = torch.cuda.Stream(device=device) if (device is not None and torch.device(device).type == "cuda") else None
        self._pending = None
        self.last_ms = None           # device time of the most recent finished gather (side stream), None until one has finished
        self._timings = []            # (start, end) events of gathers whose duration has not been read yet

    def gather(self, block):
        packed = pack_block(block)
        if self.world == 1 and not dist.is_initialized():
            return packed.unsqueeze(0)
        # concatenated-along-dim-0 output layout is accepted by both RCCL and gloo; viewed as [world, T, n, C]
        out = torch.empty((self.world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(out, packed, group=self.group)
        return out.view((self.world,) + tuple(packed.shape))

    def gather_async(self, block):
        if self._stream is None:
            self._pending = (self.gather(block), None)
            return
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)                 # the block must be complete before it is packed
        with torch.cuda.stream(self._stream):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(self._stream)
            out = self.gather(block)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self._stream)
        self._timings.append((ev0, ev))
        for t in block.values():
            t.record_stream(self._stream)
        self._pending = (out, ev)

    def wait(self):
        if self._pending is None:
            return None
        out, ev = self._pending
        self._pending = None
        if ev is not None:
            # the host runs ahead of the device: read the durations of whichever earlier gathers have finished by now, without blocking
            # (the latest one steers the caller's kernel choice)
            while self._timings and self._timings[0][1].query():
                t = self._timings.pop(0)
                self.last_ms = t[0].elapsed_time(t[1])
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # `out` was allocated on the side stream and is consumed on the caller's: tell the caching allocator, or the block could
            # be handed to the next gather while the caller's kernels still read it
            out.record_stream(cur)
        return out


class P2PRolloutGather(RolloutGather):
    """placeholder until the peer-to-peer implementation lands (same interface)"""
