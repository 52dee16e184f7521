"""MI355X-native batched simulator of the reference's `Ultrasound` environment (hot path only).

    env = UltrasoundVecEnv(4096, device="cuda:0", seed=3, **robosuite_block_of_rl_config_yaml)
    obs = env.reset(); obs, rew, done, infos = env.step(actions)        # stable-baselines3 VecEnv protocol
    obs_t, rew_t, done_t = env.step_tensor(actions_t)                    # torch tensors, no host sync

The compute path is libusim.so (HIP kernels for gfx950 behind the C ABI of include/usim.h); importing this
package without it raises."""
from . import _lib
from .config import default_robosuite_kwargs, load_yaml, make_config
from .spaces import Box
from .vec_env import UltrasoundEnv, UltrasoundVecEnv

from . import episode_log, error_metrics, policy  # noqa: E402  (checkpoint readers, on-device VecNormalize, MLP policy replay, CSV dump, error metrics)

__all__ = ["UltrasoundVecEnv", "UltrasoundEnv", "Box", "default_robosuite_kwargs", "load_yaml", "make_config", "policy", "episode_log", "error_metrics", "_lib"]
