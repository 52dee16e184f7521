"""Independent checker for the contact solver (test infrastructure): reads the dual cone QP of an environment's forward pass from the oracle's study hook
(uso_debug_dual) and solves it to machine precision with a method that shares nothing with the product's iteration -- accelerated projected gradient with
restarts and the closed-form Euclidean projection onto the friction cone.  The optimum of this strictly convex problem is what MuJoCo's Newton solver
converges to [RESTATED: MuJoCo documentation, "Computation / Solver"]."""
import ctypes as C

import numpy as np

from oracle_lib import _ptr

MAXC, ROW = 8, 10
SIZE = 2 + 36 + MAXC * 3 * ROW + MAXC * MAXC


def dual_problem(o, i, act):
    """min 1/2 f'(A + R) f + b'f over f_c in {|f_t| <= mu_c f_n}: dict with Q = A + R, b, mu (one per contact), the row Jacobians W (3 nc x 6, site space)"""
    out = np.zeros(SIZE)
    o.lib.uso_debug_dual.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    a = np.ascontiguousarray(act, dtype=np.float64)
    nc = o.lib.uso_debug_dual(o.h, i, _ptr(a), _ptr(out))
    if nc <= 0:
        return None
    Li = out[2:38].reshape(6, 6)
    rows = out[38:38 + MAXC * 3 * ROW].reshape(MAXC * 3, ROW)[:3 * nc]
    Lm = out[38 + MAXC * 3 * ROW:].reshape(MAXC, MAXC)[:nc, :nc]
    W, g, R, b = rows[:, :6], rows[:, 6], rows[:, 7], rows[:, 8]
    G = np.zeros((3 * nc, nc))
    for c in range(nc):
        G[3 * c:3 * c + 3, c] = g[3 * c:3 * c + 3]
    Aarm, Alat = W @ Li @ W.T, G @ Lm @ G.T                      # the arm's part (rank <= 6, couples every pair) and the lattice's (element to element)
    A = Aarm + Alat
    if o.cfg.probe_geoms == 2 and o.cfg.pair_model:
        # the two coincident contacts of every probe-element pair (ultrasound_probe_gripper.xml:8-9) as two contacts: the same rows twice, each with a single
        # contact's regulariser; virtual contact v < nc is contact A of pair v (cone mu_A = the environment's friction word), v >= nc contact B of pair v - nc
        muB = max(o.cfg.probe_friction2, o.cfg.elem_friction)
        A2, R2 = np.block([[A, A], [A, A]]), np.concatenate([R, R])
        return {"nc": 2 * nc, "pairs": nc, "mu": np.concatenate([np.full(nc, out[1]), np.full(nc, muB)]), "W": np.vstack([W, W]), "A": A2, "R": R2,
                "Q": A2 + np.diag(R2), "b": np.concatenate([b, b]), "Aarm": Aarm, "Alat": Alat, "Li": Li, "W1": W}
    return {"nc": nc, "pairs": 0, "mu": np.full(nc, out[1]), "W": W, "A": A, "R": R, "Q": A + np.diag(R), "b": b, "Aarm": Aarm, "Alat": Alat, "Li": Li, "W1": W}


def project_cone(f, mus):
    f = f.copy()
    mus = np.broadcast_to(np.asarray(mus, dtype=float), (len(f) // 3,))
    for c in range(len(f) // 3):
        mu = mus[c]
        n, t = f[3 * c], f[3 * c + 1:3 * c + 3]
        tn = np.linalg.norm(t)
        if tn <= mu * n:
            continue
        if mu * tn <= -n:
            f[3 * c:3 * c + 3] = 0
            continue
        nn = (n + mu * tn) / (1 + mu * mu)
        f[3 * c] = nn
        f[3 * c + 1:3 * c + 3] = t * (mu * nn / tn)
    return f


def solve_exact(P, iters=200000, tol=1e-13):
    Q, b, mu = P["Q"], P["b"], P["mu"]
    L = np.linalg.eigvalsh(Q)[-1]
    f = np.zeros_like(b); y = f.copy(); t = 1.0
    for it in range(iters):
        fn = project_cone(y - (Q @ y + b) / L, mu)
        if (fn - f) @ (y - fn) > 0:
            t = 1.0; y = fn.copy()
        else:
            tn = 0.5 * (1 + np.sqrt(1 + 4 * t * t)); y = fn + (t - 1) / tn * (fn - f); t = tn
        done = np.abs(fn - f).max() < tol and it > 50
        f = fn
        if done:
            break
    return f


def kkt_residual(P, f):
    """|f - Proj_K(f - grad)|: zero exactly at the optimum"""
    return np.abs(f - project_cone(f - (P["Q"] @ f + P["b"]), P["mu"])).max()


def net_force(P, f):
    return (P["W"].T @ f)[:3]


def primal_force(y, R, mu):
    """MuJoCo's primal view of one elliptic-cone contact [RESTATED: MuJoCo documentation, "Computation / Friction cones"]: the force as a function of the
    constraint-space acceleration y = J a - a_ref, f(y) = argmin_{f in K} 1/2 f'R f + f'y, in closed form by the three zones of the cone in R-scaled
    coordinates -- top (f = 0), bottom (inside the cone: f = -R^-1 y), middle (on its surface).  R = (Rn, Rt, Rt).  Returns (force, zone)."""
    Rn, Rt = R[0], R[1]
    mt = mu * np.sqrt(Rt / Rn)                                  # cone slope in the scaled coordinates
    yn, yt = -y[0] / np.sqrt(Rn), -y[1:] / np.sqrt(Rt)
    T = np.linalg.norm(yt)
    if T <= mt * yn:
        return np.array([yn / np.sqrt(Rn), yt[0] / np.sqrt(Rt), yt[1] / np.sqrt(Rt)]), "bottom"
    if mt * T <= -yn:
        return np.zeros(3), "top"
    p = (yn + mt * T) / (1 + mt * mt)
    e = yt / T
    return np.array([p, mt * p * e[0], mt * p * e[1]]) / np.sqrt(R), "middle"
