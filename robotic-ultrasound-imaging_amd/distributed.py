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

    def close(self):
        pass



class P2PRolloutGather:
    """The same gather without a collective kernel: every rank copies its packed block into slot `rank` of every peer's receive buffer with
    plain device-to-device copies into IPC-mapped memory (hipMemcpy between peers: the copy engines over xGMI), so no workgroups sit on the
    CUs next to the step kernels -- the split kernel keeps every SIMD ("N > 1").  Interface of RolloutGather:
    gather_async(block) right after the block has been enqueued, wait() before its consumers are enqueued; the tensor wait() returns
    ([world, T, n_local, C], a view of this rank's receive buffer) stays valid until the next gather_async() call but one.

    Protocol (DEPTH = 3 receive buffers per rank, round g uses buffer g % 3):
      set-up   every rank allocates its receive buffers, exports them (torch.multiprocessing's CUDA-IPC reduction: hipIpcGetMemHandle) and opens
               those of its peers; handles travel through `ctrl_group` (any backend; a gloo group is created when none is given).
      round g  side stream: wait for the caller's stream (the block is complete), pack, one copy per peer into peer.recv[g % 3][rank].
      wait()   host: synchronise the side stream (my copies have landed), then a barrier on ctrl_group (everybody's have).  Nothing else
               orders the ranks: a buffer is rewritten three rounds later, after two such barriers, each of which is passed only when
               every rank's block g - 1 -- enqueued behind the consumers of round g - 3 -- has finished on its GPU.
    Not yet run across real peers (the pool has one GPU per box): validated with two processes sharing one GPU (tests/test_gpu_p2p_gather.py)
    and against RolloutGather's result."""
    DEPTH = 3

    def __init__(self, group=None, device=None, ctrl_group=None):
        if not dist.is_initialized():
            raise RuntimeError("P2PRolloutGather needs an initialised torch.distributed process group (for the handle exchange and barriers)")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("P2PRolloutGather moves device memory between GPUs; use RolloutGather on the CPU")
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.ctrl = ctrl_group if ctrl_group is not None else (group if dist.get_backend(group) == "gloo" else dist.new_group(backend="gloo"))
        self._stream = torch.cuda.Stream(device=self.device)
        self._recv = None            # [DEPTH, world, T, n, C] on this rank's device
        self._peer = None            # _peer[p]: rank p's receive buffers, opened in this process
        self._round = 0
        self._pending = None
        self.last_ms = None
        self._timings = []

    def _setup(self, shape, dtype):
        from torch.multiprocessing.reductions import reduce_tensor
        with torch.cuda.device(self.device):
            self._recv = torch.zeros((self.DEPTH, self.world) + tuple(shape), dtype=dtype, device=self.device)
        torch.cuda.synchronize(self.device)
        fn, args = reduce_tensor(self._recv)
        handles = [None] * self.world
        dist.all_gather_object(handles, (fn, args), group=self.ctrl)
        self._peer = [self._recv if p == self.rank else handles[p][0](*handles[p][1]) for p in range(self.world)]
        dist.barrier(group=self.ctrl)                 # every peer has opened every buffer before the exporting tensors can go away
        self._shape = tuple(shape)

    def gather_async(self, block):
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)                 # the block must be complete before it is packed
        with torch.cuda.stream(self._stream):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(self._stream)
            packed = pack_block(block)
            if self._recv is None:
                self._setup(packed.shape, packed.dtype)
            if tuple(packed.shape) != self._shape:
                raise ValueError(f"block shape changed: {tuple(packed.shape)} vs {self._shape} (allocate one gather per block shape)")
            slot = self._round % self.DEPTH
            for k in range(self.world):               # start with the right-hand neighbour: the ranks do not all write to rank 0 first
                p = (self.rank + k) % self.world
                self._peer[p][slot, self.rank].copy_(packed, non_blocking=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(self._stream)
        self._timings.append((ev0, ev1))
        for t in block.values():
            t.record_stream(self._stream)
        packed.record_stream(self._stream)
        self._pending = slot
        self._round += 1

    def wait(self):
        if self._pending is None:
            return None
        slot, self._pending = self._pending, None
        self._stream.synchronize()                    # my copies have landed in every peer's buffer ...
        dist.barrier(group=self.ctrl)                 # ... and so have everybody else's in mine
        while self._timings and self._timings[0][1].query():
            t = self._timings.pop(0)
            self.last_ms = t[0].elapsed_time(t[1])
        return self._recv[slot]

    def gather(self, block):
        self.gather_async(block)
        return self.wait()

    def close(self):
        """Release the peers' buffers before this rank's own go away (CUDA-IPC reference counting: a producer that exits while a consumer
        still maps its memory is reported by the allocator)."""
        if self._pending is not None:
            self.wait()
        if self._peer is not None:
            self._peer = None
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.ctrl)             # nobody maps anybody's buffers any more
            torch.cuda.ipc_collect()
            self._recv = None
