"""Stand-in stepper for tests/test_bench_launch.py (USIM_BENCH_STUB=bench_stub): the surface of UltrasoundVecEnv that bench.py drives, on CPU
tensors, with transitions that encode WHICH global environment and step wrote them -- so that the test can tell that every rank simulated its own
shard (env_offset = rank * n) and that the gathered block holds all of them.  No physics; never a measurement."""
import torch


class StubEnv:
    steps_per_launch = 256

    def __init__(self, n, env_offset=0):
        self.num_envs, self.env_offset, self.action_dim = int(n), int(env_offset), 6
        self.launches = 0

    def alloc_block(self, nsteps, with_actions=True):
        n, T = self.num_envs, int(nsteps)
        blk = {"obs": torch.zeros((T, n, 19)), "rew": torch.zeros((T, n)), "done": torch.zeros((T, n), dtype=torch.uint8)}
        if with_actions:
            blk["act"] = torch.zeros((T, n, self.action_dim))
        return blk

    def block_io(self, block):
        return block                                           # the product hands a usim_step_io of raw pointers; here the tensors themselves

    def set_steps_per_launch(self, steps):
        self.steps_per_launch = int(steps)

    def reset_tensor(self):
        return torch.zeros((self.num_envs, 19))

    def refill_time(self):
        return 0.0, 0

    def rollout_random(self, first_step, nsteps, block=None, io=None):
        blk = io if io is not None else block
        self.launches += 1
        if blk is None:
            return
        gid = torch.arange(self.env_offset, self.env_offset + self.num_envs, dtype=torch.float32)
        for k in range(int(nsteps)):
            blk["obs"][k, :, 0] = gid                          # global environment id
            blk["obs"][k, :, 1] = float(first_step + k)        # global step
            blk["rew"][k] = gid * 1000.0 + float(first_step + k)
            blk["done"][k] = ((first_step + k) % 5 == 4)
            if "act" in blk:
                blk["act"][k] = 0.5

    def close(self):
        pass
