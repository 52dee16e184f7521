"""Host-side mirror of the reference's environment interface on top of libusim.

`UltrasoundVecEnv` duck-types stable_baselines3.common.vec_env.VecEnv so that it can stand where
`SubprocVecEnv([make_robosuite_env(...) for i in range(num_cpu)])` stands in src/rl.py:130 (each worker there is
Monitor(GymWrapper(suite.make("Ultrasound", **options))), src/rl.py:36-40).  `UltrasoundEnv` is the single-env
gym-style view used by src/main.py:59-70.  gym / stable-baselines3 / robosuite are not imported.

All device buffers are PyTorch-ROCm tensors; the C ABI receives their raw pointers and the current torch stream.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from .config import default_robosuite_kwargs, make_config
from .spaces import Box

_ACTION_BOX = {  # SURVEY.md Appendix D.1 (decoded from the reference checkpoints) + robosuite OSC "fixed"
    0: ([0.0] * 6, [1.0] * 6),
    1: ([-1.0] * 6, [1.0] * 6),
    2: ([0.0] * 6 + [-1.0], [1.0] * 7),
    3: ([-10.0] * 6, [10.0] * 6),
}


def _dev_index(device):
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError("UltrasoundVecEnv runs on an AMD GPU only (device must be 'cuda:N'); there is no CPU path")
    return d.index if d.index is not None else torch.cuda.current_device()


class UltrasoundVecEnv:
    """n batched `Ultrasound` environments on one MI355X.

    Parameters mirror src/rl.py:27-43: `num_envs` replaces num_cpu, `seed` is rl_config.yaml:1 (env i uses the
    stream keyed (seed, env_offset + i), the analogue of env.seed(seed + rank)), remaining kwargs are the
    `robosuite:` block of rl_config.yaml forwarded verbatim (src/rl.py:91-92)."""

    metadata = {"render.modes": []}

    def __init__(self, num_envs, device="cuda:0", seed=3, env_offset=0, monitor=True, report_truncation=False, **robosuite_kwargs):
        self.lib = _lib.load()
        if not robosuite_kwargs:
            robosuite_kwargs = default_robosuite_kwargs()
        self._kwargs = dict(robosuite_kwargs)
        self.num_envs = int(num_envs)
        self._dev = _dev_index(device)
        self.device = torch.device("cuda", self._dev)
        self._monitor = monitor
        # The reference stack (robosuite GymWrapper + Monitor, src/rl.py:36-40) never emits "TimeLimit.truncated", so SB3 does not
        # bootstrap horizon-end rewards there.  Opt in to get the key (gym TimeLimit convention) for other training set-ups.
        self._report_truncation = bool(report_truncation)
        self._env_offset = int(env_offset)
        self._handle = C.c_void_p()
        self._create(seed)
        lo, hi = _ACTION_BOX[self.cfg.mode]
        self.action_space = Box(np.array(lo), np.array(hi))
        self.observation_space = Box(np.full(_lib.OBS_DIM, -np.inf), np.full(_lib.OBS_DIM, np.inf))
        n = self.num_envs
        with torch.cuda.device(self.device):
            self._act = torch.zeros((n, self.action_dim), dtype=torch.float32, device=self.device)
            self._obs = torch.zeros((n, _lib.OBS_DIM), dtype=torch.float32, device=self.device)
            self._rew = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._done = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._term = torch.zeros((n, _lib.OBS_DIM), dtype=torch.float32, device=self.device)
            self._contacts = torch.zeros((n, 1 + _lib.MAXC), dtype=torch.int32, device=self.device)
            self._ep_ret = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._ep_len = torch.zeros(n, dtype=torch.int32, device=self.device)
            self._status = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._io = _lib.UsimStepIO(self._act.data_ptr(), self._obs.data_ptr(), self._rew.data_ptr(), self._done.data_ptr(),
                                   self._term.data_ptr(), self._contacts.data_ptr(), self._ep_ret.data_ptr(), self._ep_len.data_ptr(), None, self._status.data_ptr(), None)
        self._t_start = time.time()
        self._pending = False
        self.horizon = int(self.cfg.horizon)

    @property
    def env_offset(self):
        """global index of this shard's environment 0: every per-environment stream of the library -- resets, synthetic actions, the policy's
        exploration noise (usim_policy_step) -- is keyed (seed, env_offset + i)"""
        return self._env_offset

    # ---- construction / teardown -------------------------------------------------------------------------------
    def _create(self, seed):
        if self._handle:
            self.lib.usim_destroy(self._handle)
            self._handle = C.c_void_p()
        self.cfg = make_config(seed=seed, env_offset=self._env_offset, **self._kwargs)
        rc = self.lib.usim_create(C.byref(self.cfg), self.num_envs, self._dev, C.byref(self._handle))
        if rc != 0:
            h, self._handle = self._handle, C.c_void_p()
            try:
                _lib.check(self.lib, rc, h)
            finally:
                if h:
                    self.lib.usim_destroy(h)
        self.action_dim = self.lib.usim_action_dim(self._handle)
        self.num_elements = self.lib.usim_num_elements(self._handle)
        self.steps_per_launch = int(self.lib.usim_get_steps_per_launch(self._handle))      # the library's value (default, or USIM_STEPS_PER_LAUNCH)
        self._seed = int(seed)

    def close(self):
        if getattr(self, "_handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.usim_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        _lib.check(self.lib, rc, self._handle)

    # ---- torch-native fast path (no host synchronisation) ------------------------------------------------------
    def reset_tensor(self, mask=None):
        """Reset all envs (or those where mask != 0); returns the [n,19] observation tensor (device)."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._check(self.lib.usim_reset(self._handle, None if m is None else m.data_ptr(), self._obs.data_ptr(), self._stream()))
        return self._obs

    def reset_explicit_tensor(self, params, mask=None):
        """Reset with explicit draws, params [n,13] = start xyz, end xyz, u0, noise xyz, stiffness, damping, friction."""
        p = torch.as_tensor(params, dtype=torch.float32, device=self.device).reshape(self.num_envs, _lib.RESET_PARAMS).contiguous()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._check(self.lib.usim_reset_explicit(self._handle, None if m is None else m.data_ptr(), p.data_ptr(), self._obs.data_ptr(), self._stream()))
        return self._obs

    def step_tensor(self, actions, auto_reset=True):
        """actions: float32 [n, A] tensor on this device.  Returns (obs, rew, done) device tensors that are
        overwritten by the next call; terminal observations / episode stats are in .terminal_obs, .episode_return,
        .episode_length (valid where done)."""
        a = actions
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(a.shape) != (self.num_envs, self.action_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a.shape)}")
        self._io.act_dev = a.data_ptr()
        self._check(self.lib.usim_step(self._handle, C.byref(self._io), int(auto_reset), self._stream()))
        self._last_act = a           # keep the tensor alive until the kernel has run
        return self._obs, self._rew, self._done

    def refill_time(self):
        """(total device milliseconds, number) of the reset-bank refill launches issued so far (include/usim.h usim_refill_time)"""
        ms, cnt = C.c_double(0.0), C.c_longlong(0)
        self._check(self.lib.usim_refill_time(self._handle, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def refill_bank(self):
        """refill the reset bank now and restart its 256-step period (include/usim.h usim_refill_bank; capture-safe)"""
        self._check(self.lib.usim_refill_bank(self._handle, self._stream()))

    def set_steps_per_launch(self, steps):
        """rollout_random / time_steps: consecutive steps per kernel launch (1 .. 256, default 256; include/usim.h usim_set_steps_per_launch)"""
        self._check(self.lib.usim_set_steps_per_launch(self._handle, int(steps)))
        self.steps_per_launch = int(steps)

    def set_mapping(self, lanes_per_env, waves_per_simd=0):
        """Switch a live soft-torso env between the split kernel (lanes_per_env 32: 16-lane groups, 64: 8-lane groups) and the single-wave 16-lane kernel (waves_per_simd
        0 / 1 / 2).  The mappings compute the same bits; the choice only matters for speed (include/usim.h usim_set_mapping)."""
        self._check(self.lib.usim_set_mapping(self._handle, int(lanes_per_env), int(waves_per_simd)))

    def random_actions_tensor(self, step, out=None):
        out = self._act if out is None else out
        self._check(self.lib.usim_random_actions(self._handle, int(step), out.data_ptr(), self._stream()))
        return out

    def _block_io(self, block):
        """usim_step_io over a rollout block (dict of [T, n, ...] device tensors: obs, rew, done and optionally act)."""
        act = block.get("act")
        return _lib.UsimStepIO(None, block["obs"].data_ptr(), block["rew"].data_ptr(), block["done"].data_ptr(), None, None, None, None,
                               None if act is None else act.data_ptr(), None, None)

    def block_io(self, block):
        """The usim_step_io of a rollout block, built once and handed to rollout_random(io=...) (keeps the Python work out of a timed loop)."""
        return self._block_io(block)

    def rollout_random(self, first_step, nsteps, block=None, io=None):
        """Enqueue nsteps steps with in-kernel synthetic actions.  With `block` (or its prepared `io`), step k
        writes slice k of the [nsteps, n, ...] tensors (the transition block that is all-gathered across GPUs)."""
        advance = block is not None or io is not None
        if io is None:
            io = self._io if block is None else self._block_io(block)
        self._check(self.lib.usim_rollout_random(self._handle, int(first_step), int(nsteps), C.byref(io), int(advance), self._stream()))

    def time_steps(self, first_step, nsteps, block=None):
        """Same as rollout_random but bracketed by HIP events on the current stream; returns elapsed ms."""
        ms = C.c_float(0)
        io = self._io if block is None else self._block_io(block)
        self._check(self.lib.usim_time_steps(self._handle, int(first_step), int(nsteps), C.byref(io), int(block is not None), self._stream(), C.byref(ms)))
        return float(ms.value)

    # label of the interval that ENDS at stamp k
    # (16-lane kernel: stamps 2 / 3 / 4 close kinematics + dynamics, M^-1 + operational space, controller torque)
    PHASES = ["_", "load state", "action+fk+dynamics", "factor/inverse + op. space", "controller torque", "lattice stage+rhs", "lattice solve (MFMA)",
              "collision", "site accel + Lambda^-1", "contact rows", "pgs", "wrench", "lattice integrate", "arm acc+sensor", "arm integrate",
              "obs+reward+done", "store"]

    def profile_step(self, step):
        """Diagnostics: shader-clock ticks spent in each phase of one step kernel by wave 0 of workgroup 0 (profiling build of the
        library only: `make -C csrc prof`, USIM_LIB=.../libusim_prof.so)."""
        ticks = (C.c_uint64 * 17)()
        self._check(self.lib.usim_profile_step(self._handle, C.byref(self._io), int(step), ticks, 17))
        t = list(ticks)
        out, prev = {}, t[0]
        for k in range(1, 17):
            if t[k] and prev:
                out[self.PHASES[k]] = t[k] - prev
            if t[k]:
                prev = t[k]
        return out

    def profile_step_raw(self, step, n=64):
        """Diagnostics (profiling build): the raw shader-clock stamps of one step (see usim_profile_step in csrc/usim_api.hip)."""
        ticks = (C.c_uint64 * n)()
        self._check(self.lib.usim_profile_step(self._handle, C.byref(self._io), int(step), ticks, n))
        return list(ticks)

    def alloc_block(self, nsteps, with_actions=True):
        """Device tensors of one rollout block: obs [T,n,19], act [T,n,A], rew [T,n], done [T,n] (uint8)."""
        n, T = self.num_envs, int(nsteps)
        blk = {"obs": torch.zeros((T, n, _lib.OBS_DIM), dtype=torch.float32, device=self.device),
               "rew": torch.zeros((T, n), dtype=torch.float32, device=self.device),
               "done": torch.zeros((T, n), dtype=torch.uint8, device=self.device)}
        if with_actions:
            blk["act"] = torch.zeros((T, n, self.action_dim), dtype=torch.float32, device=self.device)
        return blk

    def enable_step_log(self, enable=True):
        """Allocate (or drop) the [n, 53] per-step episode record the step kernel fills (channels of the reference's save_data dump)."""
        if enable:
            self._log = torch.zeros((self.num_envs, _lib.LOG_WIDTH), dtype=torch.float32, device=self.device)
            self._io.log_dev = self._log.data_ptr()
        else:
            self._log = None
            self._io.log_dev = None
        return getattr(self, "_log", None)

    @property
    def step_log(self):
        return getattr(self, "_log", None)

    @property
    def status(self):
        """int32 [n] status word of the last step (bit 0: contact-slot overflow, bit 1: full torso, element-table contact overflow, bit 2: numerical fault -> episode ended)"""
        return self._status

    @property
    def terminal_obs(self):
        return self._term

    @property
    def contacts(self):
        return self._contacts

    @property
    def episode_return(self):
        return self._ep_ret

    @property
    def episode_length(self):
        return self._ep_len

    # ---- stable-baselines3 VecEnv protocol (numpy in / numpy out) --------------------------------------------------
    def reset(self):
        obs = self.reset_tensor()
        return obs.cpu().numpy().copy()

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32)
        if a.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {a.shape}")
        self._act.copy_(torch.from_numpy(np.ascontiguousarray(a)), non_blocking=False)
        self.step_tensor(self._act)
        self._pending = True

    def step_wait(self):
        if not self._pending:
            raise RuntimeError("step_wait called without step_async")
        self._pending = False
        obs = self._obs.cpu().numpy().copy()
        rew = self._rew.cpu().numpy().copy()
        done = self._done.cpu().numpy().astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        if done.any():
            idx = np.nonzero(done)[0]
            term = self._term.cpu().numpy()
            ep_r = self._ep_ret.cpu().numpy()
            ep_l = self._ep_len.cpu().numpy()
            now = round(time.time() - self._t_start, 6)
            for i in idx:
                infos[i]["terminal_observation"] = term[i].copy()
                if self._monitor:                                # SB3 Monitor (src/rl.py:39)
                    infos[i]["episode"] = {"r": float(ep_r[i]), "l": int(ep_l[i]), "t": now}
                if self._report_truncation:
                    infos[i]["TimeLimit.truncated"] = bool(ep_l[i] >= self.horizon)
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        """VecEnv.seed: re-keys the per-env streams (env i <- (seed, env_offset + i)); call reset() afterwards."""
        s = self._seed if seed is None else int(seed)
        torch.cuda.synchronize(self.device)
        self._create(s)
        return [s + self._env_offset + i for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError("per-environment attributes are fixed at construction in the batched simulator")

    def env_method(self, method_name, *args, indices=None, **kwargs):
        fn = getattr(self, method_name)
        return [fn(*args, **kwargs) for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        raise NotImplementedError("rendering is outside the simulated hot path")

    def render(self, mode="human"):
        return None

    @property
    def unwrapped(self):
        return self

    # ---- checkpoint / inspection ----------------------------------------------------------------------------------
    _FIELDS = {"q": slice(0, 7), "qd": slice(7, 14), "q0": slice(14, 21), "traj_start": slice(21, 24), "traj_end": slice(24, 27),
               "u0": 27, "vbar": 28, "fzbar": 29, "fzprev": 30, "dfz": 31, "stiffness": 32, "damping": 33, "mu": 34, "t": 35,
               "has_touched": 36, "episode": 37, "ep_return": 38, "status": 39}

    def get_state(self):
        sc = np.zeros((self.num_envs, _lib.NSCALAR), dtype=np.float32)
        lat = np.zeros((self.num_envs, max(self.num_elements, 1), 2), dtype=np.float32)
        self._check(self.lib.usim_get_state(self._handle, sc.ctypes.data, lat.ctypes.data))
        st = {k: sc[:, v].copy() for k, v in self._FIELDS.items()}
        st["s"] = lat[:, : self.num_elements, 0].copy()
        st["sd"] = lat[:, : self.num_elements, 1].copy()
        if self.num_elements == 270:                                  # full torso: the free body (position world, quaternion w x y z, velocity world, angular velocity body frame)
            body = np.zeros((self.num_envs, 13 + 4 * 270 + 8 + 64), dtype=np.float64)             # USIM_FULL_BODY_WORDS
            self._check(self.lib.usim_get_body_state(self._handle, body.ctypes.data))
            st["body"] = body[:, :13].copy()
            st["solver_warm_start"] = body[:, 13:].copy()                                          # the contact forces of the previous physics step (the solve's initial guess)
        return st

    def set_state(self, st):
        sc = np.zeros((self.num_envs, _lib.NSCALAR), dtype=np.float32)
        lat = np.zeros((self.num_envs, max(self.num_elements, 1), 2), dtype=np.float32)
        for k, v in self._FIELDS.items():
            sc[:, v] = np.asarray(st[k], dtype=np.float32)
        if self.num_elements:
            lat[:, : self.num_elements, 0] = st["s"]
            lat[:, : self.num_elements, 1] = st["sd"]
        if self.num_elements == 270:
            body = np.ascontiguousarray(np.concatenate([np.asarray(st["body"], dtype=np.float64), np.asarray(st["solver_warm_start"], dtype=np.float64)], axis=1))
            self._check(self.lib.usim_set_body_state(self._handle, body.ctypes.data))
        self._check(self.lib.usim_set_state(self._handle, sc.ctypes.data, lat.ctypes.data))


class UltrasoundEnv:
    """Single-environment gym-style view (src/main.py:59-70): reset() -> obs, step(a) -> (obs, r, done, info).
    No auto-reset: stepping a finished episode raises ValueError like robosuite's MujocoEnv.step."""

    def __init__(self, device="cuda:0", seed=3, **robosuite_kwargs):
        save_data = bool(robosuite_kwargs.get("save_data", False))
        self._vec = UltrasoundVecEnv(1, device=device, seed=seed, monitor=False, **robosuite_kwargs)
        self._logger = None
        if save_data:                                   # ultrasound.py:479-509: per-episode CSV dump of the single environment
            from .episode_log import EpisodeLogger
            self._logger = EpisodeLogger(self._vec, 0, root=".")
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.horizon = self._vec.horizon
        self.action_dim = self._vec.action_dim
        self.done = True
        self._last_reward = 0.0

    @property
    def action_spec(self):
        return self.action_space.low.copy(), self.action_space.high.copy()

    def seed(self, seed=None):
        self._vec.seed(seed)
        self.done = True

    def reset(self):
        self.done = False
        return self._vec.reset()[0]

    def step(self, action):
        if self.done:
            raise ValueError("executing action in terminated episode")
        a = torch.as_tensor(np.asarray(action, dtype=np.float32)).reshape(1, self.action_dim).to(self._vec.device)
        obs, rew, done = self._vec.step_tensor(a, auto_reset=False)
        torch.cuda.synchronize(self._vec.device)
        self.done = bool(done[0].item())
        self._last_reward = float(rew[0].item())
        if self._logger is not None:
            self._logger.after_step(self.done)
        return obs[0].cpu().numpy().copy(), self._last_reward, self.done, {}

    def reward(self, action=None):
        return self._last_reward

    def close(self):
        self._vec.close()
