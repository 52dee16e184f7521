"""Per-environment state differences GPU (float32) vs oracle (float64) after 200 steps, for every mode, and for the float32 BUILD of the oracle against the float64 one
(what precision alone does): the data behind the bars of tests/test_gpu_parity.py.   python tests/studies/parity_report.py [n]   (GPU box)"""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_parity as T
from oracle_lib import Oracle
usim = importlib.import_module("robotic-ultrasound-imaging_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["tracking", "fixed", "variable_z", "wrench"]
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
for extra in (({},) if n > 1024 else ({}, {"probe_geoms": 1})):
    for mode in modes:
        T.REPORT = {}
        expl, dead = T._run_parity(usim, n, 200, "soft", mode, omp=n > 1024, **extra)
        r = T.REPORT
        line = " ".join(f"{k}: q50 {np.median(r[k]):.1e} q99 {np.quantile(r[k], .99):.1e} max {r[k].max():.1e} (#>1e-4: {(r[k] > 1e-4).sum()}; scale {r[k + '_scale']:.2g})" for k in ("q", "qd", "s", "sd"))
        print(f"GPU  {mode:10s} {extra} razor {dead}/{n} | {line}", flush=True)
        # float32 oracle vs float64 oracle on the same run
        if n > 1024:
            continue
        a, b = Oracle(n, precision="f64", mode=mode, **extra), Oracle(n, precision="f32", mode=mode, **extra)
        a.reset(); b.reset(); alive = np.ones(n, bool)
        for k in range(200):
            act = a.random_actions(k); ra, rb = a.step(act), b.step(act)
            alive &= (ra[2] == rb[2]) & (ra[4] == rb[4]).all(1)
        sa, sb = a.get_state(), b.get_state()
        out = []
        for key in ("q", "qd", "s", "sd"):
            pe = np.abs(sa[key][alive] - sb[key][alive]).reshape(alive.sum(), -1).max(1) / np.abs(sa[key][alive]).max()
            out.append(f"{key}: q99 {np.quantile(pe, .99):.1e} max {pe.max():.1e} (#>1e-4: {(pe > 1e-4).sum()})")
        print(f"f32o {mode:10s} {extra} diverged {(~alive).sum()}/{n} | " + " ".join(out), flush=True)
