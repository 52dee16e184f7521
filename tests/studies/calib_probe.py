#!/usr/bin/env python3
"""Calibration of the probe stand-in (the probe mesh is missing from the reference snapshot) against the 192 decoded
reset observations of the reference checkpoints (tests/golden/reference_pins.npz).

    python tests/studies/calib_probe.py                                   # reference statistics vs the current defaults
    python tests/studies/calib_probe.py probe_radius=0.01,0.012 probe_height=0.04,0.045    # sweep (cartesian product)
    python tests/studies/calib_probe.py --torque-fit                      # fixed-lever-arm fit of the reference torques

Runs the CPU oracle only (test infrastructure, which is why this script lives under tests/; it is not collected by pytest); the kernels use the
same geometry (tests/test_gpu_parity.py).
Round-3 record: profiles/r03/calib_probe.txt."""
import itertools
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from oracle_lib import Oracle  # noqa: E402

np.set_printoptions(precision=2, suppress=True, linewidth=220)
PINS = np.load(ROOT / "tests" / "golden" / "reference_pins.npz")
REF = np.concatenate([PINS[f"{m}_reset_obs"] for m in ("tracking", "variable_z", "wrench")])
BINS = ((-0.03, -0.015), (-0.015, -0.005), (-0.005, 0), (0, 0.005), (0.005, 0.010), (0.010, 0.015))


def stats(O):
    c = O[:, 2] > 0.5
    F, T = O[c, 0:3], O[c, 3:6]
    b = [O[(O[:, 14] >= lo) & (O[:, 14] < hi), 2].mean() for lo, hi in BINS]
    lat = np.hypot(F[:, 0], F[:, 1]) / F[:, 2]
    return dict(Fm=F.mean(0), Fs=F.std(0), Tm=T.mean(0), Ts=T.std(0), bins=np.array(b), lat=np.median(lat),
                q99=np.quantile(np.abs(F[:, :2]), 0.99, axis=0), cf=c.mean(), cxz=np.corrcoef(F[:, 0], F[:, 2])[0, 1])


def show(v, label):
    print(label, "F mean", v["Fm"], "std", v["Fs"], "| T mean", v["Tm"], "std", v["Ts"], "| Fz by depth bin", v["bins"],
          "| median lateral/Fz %.2f  q99 |Fx|,|Fy|" % v["lat"], v["q99"], " contact frac %.2f corr(Fx,Fz) %.2f" % (v["cf"], v["cxz"]))


def torque_fit():
    """T - T0 = F x r in the site frame: is there one lever arm that explains the reference torques?  (No: 23 % / 17 % / 0 %
    of the variance -- the contact points vary from row to row, i.e. few dominant contacts.)"""
    c = REF[:, 2] > 0.5
    T0 = REF[~c, 3:6].mean(0)
    x, y, z, w = -0.69192486, 0.72186726, -0.00514253, -0.01100909
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Fs, dT = REF[c, 0:3] @ Rm, REF[c, 3:6] - T0
    A = np.concatenate([np.array([[0, -F[2], F[1]], [F[2], 0, -F[0]], [-F[1], F[0], 0]]) for F in Fs])
    r = np.linalg.lstsq(A, dT.reshape(-1), rcond=None)[0]
    pred = (A @ r).reshape(-1, 3)
    print("site axes in world (columns):\n", Rm)
    print("lever arm (site frame)", r, " explained variance per torque channel", 1 - ((dT - pred) ** 2).sum(0) / ((dT - dT.mean(0)) ** 2).sum(0))


if __name__ == "__main__":
    if "--torque-fit" in sys.argv:
        torque_fit()
        sys.exit(0)
    show(stats(REF), "REFERENCE (192 rows) ")
    names, lists = [], []
    for a in sys.argv[1:]:
        k, v = a.split("=")
        names.append(k)
        lists.append([(int(x) if x.lstrip('-').isdigit() else float(x)) for x in v.split(",")])
    for combo in itertools.product(*lists):
        kw = dict(zip(names, combo))
        o = Oracle(1536, **kw)
        show(stats(o.reset()), "oracle " + " ".join(f"{k}={v}" for k, v in kw.items()))
