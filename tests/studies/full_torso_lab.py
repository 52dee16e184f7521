"""Round-6 study (CPU, oracle + numpy): the full torso's contact solve -- ~54 sticking element-table contacts on ONE rigid body plus the probe's pairs -- converges slowly under the
warm-started Gauss-Seidel of round 5 (94 % of a converged solve's decisions at 24 sweeps, 97 % at 48).  Hypothesis (from tests/studies/pair_lab.py: the arm's coupling of sticking
contacts binds the top-face model's percentiles): the slow modes are the twelve rigid coordinates the rows share (the torso body's six, the arm's six at the site), and a COARSE
CORRECTION on them after every sweep would fix it.  Measured on the dual problems the oracle exports (uso_debug_full), against 3000 sweeps (KKT residual 2e-9):
  gs        round 5: sweeps of the continuous local solve (solver_lab.local_solve)
  gs+free   after every sweep the exact minimiser of the quadratic over f + Z z, Z = D^-1 U restricted to the contacts strictly inside their cones (U: the rows' 3 n x 12 rigid
            Jacobian, D: the diagonal blocks), step limited by the cones
  gs+proj   the same over all contacts, projected onto the cones, taken with an exact line search if it descends
Result: the correction buys a factor ~3 on the probe's net force at 24 sweeps and nothing on max |f - f*| -- the slow modes are NOT the rigid ones.  Q's smallest eigenvalues are
the friction regulariser R_t = R_n / 20 (0.09 /kg against 49 /kg at the top: condition number 550): ~100 modes that redistribute the tangential forces among the sticking table contacts
without a net force or torque on the body, curvature R_t in blocks of curvature 0.46.  They hardly move the body, the sliders or the probe (their error is in |f - f*|, not in the
outputs); what does reach the outputs converges at the Gauss-Seidel's own rate.  A converged full torso needs MuJoCo's road (Newton / CG on the primal: 6 + 6 + 270 accelerations, where
the regulariser is a large stiffness instead of a small one), not a coarse space on the dual.   usage: python tests/studies/full_torso_lab.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "studies"))
from cone_qp import project_cone          # noqa: E402
from oracle_lib import Oracle, _ptr       # noqa: E402
from solver_lab import local_solve        # noqa: E402


def problem(o, i, act):
    cap = 3 + 240 * 240 + 240 * 14 + 2 * 80
    out = np.zeros(cap)
    o.lib.uso_debug_full.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long]
    nv = o.lib.uso_debug_full(o.h, i, _ptr(np.ascontiguousarray(act, dtype=np.float64)), _ptr(out), cap)
    if nv <= 0:
        return None
    nr = 3 * nv; p = 3
    Q = out[p:p + nr * nr].reshape(nr, nr).copy(); p += nr * nr
    b = out[p:p + nr].copy(); p += nr
    mu = out[p:p + nv].copy(); p += nv
    f0 = out[p:p + nr].copy(); p += nr
    lam0 = out[p:p + nv].copy(); p += nv
    U = out[p:p + nr * 12].reshape(nr, 12).copy()
    return {"nv": nv, "nc": int(out[1]), "nt": int(out[2]), "Q": Q, "b": b, "mu": mu, "f0": f0, "lam0": lam0, "U": U}


def sweep(P, f, lam, r):
    Q, mu = P["Q"], P["mu"]
    for c in range(P["nv"]):
        i = slice(3 * c, 3 * c + 3)
        fn, lam[c] = local_solve(Q[i, i], r[i], f[i].copy(), mu[c], lam[c])
        r += Q[:, i] @ (fn - f[i]); f[i] = fn
    return f, lam, r


def run(P, sweeps, mode="gs", warm=True):
    Q, nv, mu = P["Q"], P["nv"], P["mu"]
    f = P["f0"].copy() if warm else np.zeros(3 * nv); lam = P["lam0"].copy() if warm else np.zeros(nv)
    r = Q @ f + P["b"]
    Z = np.zeros_like(P["U"])
    for c in range(nv):
        i = slice(3 * c, 3 * c + 3); Z[i] = np.linalg.solve(Q[i, i], P["U"][i])
    Z = Z[:, np.abs(Z).sum(0) > 0]
    ZQZ_inv = np.linalg.pinv(Z.T @ Q @ Z)
    hist = []
    for _ in range(sweeps):
        f, lam, r = sweep(P, f, lam, r)
        d = None
        if mode == "gs+proj":
            d = project_cone(f - Z @ (ZQZ_inv @ (Z.T @ r)), mu) - f; tmax = 1.0
        elif mode == "gs+free":
            free = (f[0::3] > 1e-9) & (np.hypot(f[1::3], f[2::3]) < mu * f[0::3] * (1 - 1e-6))
            Zf = Z * np.repeat(free, 3)[:, None]
            d = -Zf @ (np.linalg.pinv(Zf.T @ Q @ Zf) @ (Zf.T @ r)); tmax = 1.0
            for c in np.nonzero(free)[0]:
                i = slice(3 * c, 3 * c + 3)
                while tmax > 1e-6 and not ((f[i] + tmax * d[i])[0] >= 0 and np.hypot(*(f[i] + tmax * d[i])[1:]) <= mu[c] * (f[i] + tmax * d[i])[0]):
                    tmax *= 0.7
        if d is not None and r @ d < 0:
            Qd = Q @ d; t = min(tmax, -(r @ d) / (d @ Qd)); f = f + t * d; r = r + t * Qd
        hist.append(f.copy())
    return hist


if __name__ == "__main__":
    n, pre = 12, 25
    o = Oracle(n, torso="full", pgs_iters=48); o.reset()
    for k in range(pre):
        o.step(o.random_actions(k))
    act = o.random_actions(pre)
    probs = [p for p in (problem(o, i, act[i]) for i in range(n)) if p is not None]
    print(f"{len(probs)} problems, {pre} steps after a reset: virtual contacts {[p['nv'] for p in probs]} (probe pairs {[p['nc'] for p in probs]})", flush=True)
    star = [run(p, 3000)[-1] for p in probs]
    w = [np.linalg.eigvalsh(p["Q"]) for p in probs]
    print("eigenvalues of Q: smallest %.3f .. %.3f, largest %.1f .. %.1f /kg" % (min(x[0] for x in w), max(x[0] for x in w), min(x[-1] for x in w), max(x[-1] for x in w)))
    its = (1, 2, 4, 8, 12, 16, 24, 48)
    for warm in (True, False):
        for mode in ("gs", "gs+free", "gs+proj"):
            E = []; EW = []
            for p, x in zip(probs, star):
                h = run(p, max(its), mode, warm)
                E.append([np.abs(h[k - 1] - x).max() for k in its])
                EW.append([np.abs(p["U"][:, 6:9].T @ (h[k - 1] - x)).max() for k in its])           # the probe's net force: site rows of the probe contacts
            E = np.array(E); EW = np.array(EW)
            label = f"{mode}, {'warm start' if warm else 'cold'}"
            print(f"{label:22s} max |f - f*| (N) median/worst by sweeps: " + "  ".join(f"{k}: {np.median(E[:, j]):.0e}/{E[:, j].max():.0e}" for j, k in enumerate(its)))
            print(f"{'':22s} probe's net force                      : " + "  ".join(f"{k}: {np.median(EW[:, j]):.0e}/{EW[:, j].max():.0e}" for j, k in enumerate(its)), flush=True)
