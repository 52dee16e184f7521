"""The C ABI used directly, without the Python host classes: the stand-alone ctypes stub of INTEGRATION.md section 4 must work
as printed (it is extracted from the document and executed), and the error conventions must hold on a real device."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_integration_md_stub_runs_as_printed(monkeypatch):
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class UsimStepIO(C.Structure)" in b)
    monkeypatch.chdir(ROOT)                                  # the stub uses the in-tree library path
    ns = {}
    exec(compile(stub, "INTEGRATION.md:stub", "exec"), ns)
    torch.cuda.synchronize()
    obs, rew, done = ns["obs"], ns["rew"], ns["done"]
    assert obs.shape == (4096, 19) and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert (rew > 0).all() and (rew <= 12.0001).all() and done.dtype == torch.uint8


def test_error_conventions_on_device(usim):
    lib = usim._lib.load()
    cfg = usim.make_config()
    h = C.c_void_p()
    assert lib.usim_create(C.byref(cfg), 8, 99, C.byref(h)) == -2 and not h            # device index out of range
    cfg.lanes_per_env = 5
    assert lib.usim_create(C.byref(cfg), 8, 0, C.byref(h)) == -1                        # invalid mapping
    if h:
        lib.usim_destroy(h); h = C.c_void_p()
    cfg.lanes_per_env = 0
    assert lib.usim_create(C.byref(cfg), 8, 0, C.byref(h)) == 0 and h
    assert lib.usim_num_envs(h) == 8 and lib.usim_action_dim(h) == 6 and lib.usim_num_elements(h) == 99
    io = usim._lib.UsimStepIO()                                                          # all-NULL io block
    assert lib.usim_step(h, C.byref(io), 1, None) == -1
    assert lib.usim_reset_explicit(h, None, None, None, None) == -1
    assert lib.usim_last_hip_error(h).decode() == ""
    # state round trip through the raw entry points
    assert lib.usim_reset(h, None, None, None) == 0
    sc = np.zeros((8, 40), dtype=np.float32); lat = np.zeros((8, 99, 2), dtype=np.float32)
    assert lib.usim_get_state(h, sc.ctypes.data, lat.ctypes.data) == 0
    assert np.all(sc[:, 37] == 1) and np.all(sc[:, 35] == 0) and np.isfinite(sc).all()   # episode 1, t = 0
    assert lib.usim_set_state(h, sc.ctypes.data, lat.ctypes.data) == 0
    sc2 = np.zeros_like(sc)
    assert lib.usim_get_state(h, sc2.ctypes.data, None) == 0 and np.array_equal(sc, sc2)
    lib.usim_destroy(h)


def test_profile_step_needs_the_profiling_build(usim):
    """the phase stamps are compiled into libusim_prof.so only (make -C csrc prof); the production library says so instead of
    returning zeros"""
    env = usim.UltrasoundVecEnv(64, device="cuda:0", seed=1, **usim.default_robosuite_kwargs())
    env.reset_tensor()
    with pytest.raises(RuntimeError, match="(?i)unsupported|profiling build"):
        env.profile_step(0)
    obs, rew, done = env.step_tensor(env.random_actions_tensor(0))      # the handle is still usable
    assert torch.isfinite(obs).all()
    env.close()
