#!/usr/bin/env python3
"""Decode the only numeric pins the reference ships into small fixtures.

Source blobs (DATA, not code): /root/reference/src/trained_rl_models/{tracking,variable_z,wrench}.zip
and vec_normalize_{...}.pkl.  They are SB3 1.1.0a5 checkpoints; we only use json+base64+pickle+numpy
with stub classes, never SB3/gym (not installed).  Run in the authoring container only:

    python tests/golden/make_fixtures.py

Writes tests/golden/reference_pins.npz (+ reference_pins.json with scalars/metadata).
"""
import base64, io, json, pickle, sys, types, zipfile
from pathlib import Path
import numpy as np

REF = Path("/root/reference/src/trained_rl_models")
OUT = Path(__file__).resolve().parent
MODELS = ["tracking", "variable_z", "wrench"]


def _stub(modname, clsname):
    parts = modname.split(".")
    for i in range(1, len(parts) + 1):
        name = ".".join(parts[:i])
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    cls = type(clsname, (), {"__setstate__": lambda self, st: self.__dict__.update(st)})
    setattr(sys.modules[modname], clsname, cls)
    return cls


def _decode(field):
    return pickle.loads(base64.b64decode(field[":serialized:"]))


def main():
    _stub("gym.spaces.box", "Box")
    _stub("gym.spaces.space", "Space")
    _stub("stable_baselines3.common.running_mean_std", "RunningMeanStd")
    _stub("stable_baselines3.common.vec_env.vec_normalize", "VecNormalize")
    arrays, meta = {}, {}
    for m in MODELS:
        with zipfile.ZipFile(REF / f"{m}.zip") as z:
            d = json.loads(z.read("data"))
        arrays[f"{m}_reset_obs"] = np.asarray(_decode(d["_last_original_obs"]), dtype=np.float64)
        asp = _decode(d["action_space"]).__dict__
        osp = _decode(d["observation_space"]).__dict__
        arrays[f"{m}_action_low"] = np.asarray(asp["low"], dtype=np.float64)
        arrays[f"{m}_action_high"] = np.asarray(asp["high"], dtype=np.float64)
        ep = _decode(d["ep_info_buffer"])
        meta[m] = {
            "n_envs": d["n_envs"], "num_timesteps": d["num_timesteps"],
            "obs_shape": list(osp["_shape"]) if "_shape" in osp else list(osp["shape"]),
            "sb3_version": d.get("_stable_baselines3_version") or d.get("policy_class", {}).get("__module__"),
            "ep_mean_return": float(np.mean([e["r"] for e in ep])),
            "ep_mean_length": float(np.mean([e["l"] for e in ep])),
            "ep_max_wall_s": float(np.max([e["t"] for e in ep])),
        }
        with open(REF / f"vec_normalize_{m}.pkl", "rb") as f:
            vn = pickle.load(f)
        v = vn.__dict__
        arrays[f"{m}_obs_rms_mean"] = np.asarray(v["obs_rms"].__dict__["mean"], dtype=np.float64)
        arrays[f"{m}_obs_rms_var"] = np.asarray(v["obs_rms"].__dict__["var"], dtype=np.float64)
        arrays[f"{m}_old_obs"] = np.asarray(v["old_obs"], dtype=np.float64)
        meta[m].update({
            "obs_rms_count": float(v["obs_rms"].__dict__["count"]),
            "ret_rms_mean": float(v["ret_rms"].__dict__["mean"]),
            "ret_rms_var": float(v["ret_rms"].__dict__["var"]),
            "clip_obs": float(v["clip_obs"]), "clip_reward": float(v["clip_reward"]),
            "gamma": float(v["gamma"]), "epsilon": float(v["epsilon"]),
        })
    # policy weights of the `tracking` checkpoint (policy.pth inside the zip is a plain tensor state dict): data for the
    # policy-replay test; stored as float16-exact? no -- float32 as shipped
    import io, torch
    for m in MODELS:
        with zipfile.ZipFile(REF / f"{m}.zip") as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        np.savez_compressed(OUT / f"{m}_policy.npz", **{k: v.numpy() for k, v in sd.items()})
    np.savez_compressed(OUT / "reference_pins.npz", **arrays)
    (OUT / "reference_pins.json").write_text(json.dumps(meta, indent=1, sort_keys=True))
    for k, a in arrays.items():
        print(k, a.shape)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
