"""Round-6 solver study (CPU, oracle + numpy): WHY does the block Jacobi iteration of round 5 contract by ~0.5 per pass, and is there a pair-aware visit that holds the
accuracy bars at half the iterations?  The round-5 review's hypothesis: contacts A and B of a probe-element pair (ultrasound_probe_gripper.xml:8-9) have identical rows,
so the line search returns t ~ 1/2 and every weakly coupled mode contracts by ~1/2 per pass.  Measured here on the dual problems the oracle exports (uso_debug_dual)
against the independent optimum (tests/cone_qp.py):

  t        the line search's step length by iteration                                   -> median 0.8 - 1.0, not 1/2: the regulariser of a normal row (1.8 - 3.6 /kg) is
                                                                                           larger than its Delassus entry (0.1 - 2 /kg), two coincident normal rows overshoot by 20 %, not 100 %
  coupling contraction per pass with the arm's / the lattice's / all coupling BETWEEN pairs removed -> 0.50 with every pair on its own: the slow mode lives INSIDE a pair --
                                                                                           but in its TANGENTIAL rows (regulariser R_n / 20): contact A's friction (disc of mu_A f_n ~ 0.05 N) chases the
                                                                                           direction of contact B's, and the pair's tangential difference mode has curvature R_t against A_tt + R_t in each block
  model    the visit on a model block K o A_cc + R that anticipates the partner (kn = 2 ...)  -> worse at every count (the fixed point is the same; the normal rows were not the problem)
  gs       contact A's visit, then B's on the residual A left (two visits per pass)       -> 12 passes ~ 16 Jacobi iterations at twice the visits
  pair     the pair's 6 x 6 two-cone block solved (nearly) exactly per pass                 -> median contraction 0.27, but the 99th percentile -- three or more pairs whose B contacts all
                                                                                           STICK -- still contracts by ~0.63: their tangential rows are coupled through the arm (0.13 - 0.19 /kg each to each)
                                                                                           against R_t = 0.09 /kg; the review's bars (99 % / worst) are reached at 11 - 12 passes of >= 2 visits: no gain
  arm      model blocks with the arm's part scaled by the number of pairs                   -> worse
  momentum Nesterov / constant extrapolation of the point the visits are evaluated at       -> worse (the exact line search already is the one-dimensional Krylov step)
  cheap    a cheaper visit: the friction step without the multiplier's Newton step (unconstrained tangential minimiser, radial clamp), or without the first ray
                                                                                        -> rests 0.5 - 10 N from the optimum: the visit's fixed point is the optimum only with both
  newton   MuJoCo's own road: Newton on the primal (6 site + nc element accelerations), exact line search -> 5 - 6 iterations to 1e-3 / 4e-2 N, each a (6 + nc)^2 Hessian assembly,
                                                                                           a factorisation and a piecewise line search: ~4 Jacobi iterations of instructions apiece on one wave
Conclusion: no visit of comparable cost reaches the bars in <= 12 iterations; block Jacobi + line search at one visit per pass stays.
usage: python tests/studies/pair_lab.py [build|t|coupling|cands|pair|newton|all]"""
import pickle
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "studies"))
from cone_qp import dual_problem, net_force, primal_force, solve_exact   # noqa: E402
from oracle_lib import Oracle                                             # noqa: E402
from solver_lab import local_solve                                        # noqa: E402

CACHE = Path("/tmp/pair_lab_sets.pkl")


def build_sets():
    sets = {}
    for n, pre in ((256, 8), (256, 40), (512, 300)):
        o = Oracle(n, pair_model=1); o.reset()
        for k in range(pre):
            o.step(o.random_actions(k))
        act = o.random_actions(pre)
        probs = [p for p in (dual_problem(o, i, act[i]) for i in range(n)) if p is not None]
        sets[f"{n} envs, {pre} steps after a reset"] = (probs, [solve_exact(p) for p in probs])
    CACHE.write_bytes(pickle.dumps(sets))
    return sets


def load_sets():
    return pickle.loads(CACHE.read_bytes()) if CACHE.exists() else build_sets()


def local_solve_cheap(B, r, f, mu, lam, ray1=True):
    """solver_lab.local_solve with pieces left out (what does a visit need for its fixed point to be the optimum?): the friction step is the unconstrained tangential
    minimiser clamped radially (no multiplier); ray1 = False also drops the ray along the current force"""
    fc = f.copy()
    if ray1 and f[0] > 1e-10:
        Bf = B @ f; x = max(-1.0, -(f @ r) / (f @ Bf)); fc = f + x * f; r = r + x * Bf
    rtn = np.hypot(r[1], r[2])
    v = np.array([1.0, -mu * r[1] / rtn, -mu * r[2] / rtn]) if (rtn > 0 and r[0] < mu * rtn) else np.array([1.0, 0.0, 0.0])
    Bv = B @ v; x = -(v @ r) / (v @ Bv); x = max(0.0, x) if ray1 else max(x, -fc[0])
    fc = fc + x * v; r = r + x * Bv
    lim = mu * fc[0]; t = np.zeros(2)
    if lim > 1e-7:
        Btt = B[1:, 1:]; q = r[1:] - Btt @ fc[1:]
        t = -np.linalg.solve(Btt, q); tn = np.hypot(*t)
        if tn > lim:
            t = t * lim / tn
    return np.array([fc[0], t[0], t[1]]), lam


def jacobi(P, iters, K=None, arm_scale=0.0, beta=0.0, trace=None, history=False, visit=None):
    """round 5's iteration; K (3 x 3 weights on A_cc) / arm_scale give the visits a MODEL block, beta extrapolates the point the visits are evaluated at"""
    Q, b, mu, nv, R, nc = P["Q"], P["b"], P["mu"], P["nc"], P["R"], P["pairs"]
    f = np.zeros(3 * nv); lam = np.zeros(nv); p = np.zeros(3 * nv); hist = []
    Bm = []
    for v in range(nv):
        c = v % nc; i = slice(3 * c, 3 * c + 3)
        Acc = P["A"][i, i] if K is None else K * P["A"][i, i]
        Bm.append(Acc + arm_scale * (nc - 1) * P["Aarm"][i, i] + np.diag(R[i]))
    for it in range(iters):
        y = f + beta * p
        r = Q @ y + b; fh = f.copy()
        for v in range(nv):
            i = slice(3 * v, 3 * v + 3)
            fh[i], lam[v] = (visit or local_solve)(Bm[v], r[i] + Bm[v] @ (f[i] - y[i]), f[i], mu[v], lam[v])
        d = fh - f; den = d @ Q @ d
        num = (Q @ f + b) @ d if beta else -sum(d[3 * v:3 * v + 3] @ Bm[v] @ d[3 * v:3 * v + 3] for v in range(nv))
        t = max(0.0, min(1.0, -num / den)) if den > 0 else 0.0
        if trace is not None and den > 0:
            trace.append((it, t))
        p = t * d; f = f + p
        hist.append(f.copy())
    return hist if history else f


def pair_block(P, iters, inner, history=False):
    """Jacobi + line search ACROSS pairs; inside a pair `inner` rounds of (visit A, visit B on the residual A left): inner = 1 is "gs", inner = 40 an exact pair block"""
    Q, b, mu, nv, nc = P["Q"], P["b"], P["mu"], P["nc"], P["pairs"]
    f = np.zeros(3 * nv); lam = np.zeros(nv); hist = []
    for _ in range(iters):
        r = Q @ f + b; fh = f.copy()
        for c in range(nc):
            i = slice(3 * c, 3 * c + 3); j = slice(3 * (nc + c), 3 * (nc + c) + 3); A = P["A"][i, i]
            fa, fb, ra, rb = f[i].copy(), f[j].copy(), r[i].copy(), r[j].copy()
            for _k in range(inner):
                fn, lam[c] = local_solve(Q[i, i], ra, fa, mu[c], lam[c]); dd = fn - fa; fa = fn; ra = ra + Q[i, i] @ dd; rb = rb + A @ dd
                fn, lam[nc + c] = local_solve(Q[j, j], rb, fb, mu[nc + c], lam[nc + c]); dd = fn - fb; fb = fn; rb = rb + Q[j, j] @ dd; ra = ra + A @ dd
            fh[i] = fa; fh[j] = fb
        d = fh - f; den = d @ Q @ d
        f = f + (max(0.0, min(1.0, -(r @ d) / den)) if den > 0 else 0.0) * d
        hist.append(f.copy())
    return hist if history else f


def primal_newton(P, iters):
    """MuJoCo's formulation: unknowns u = (site acceleration 6, element accelerations nc), cost 1/2 u'H u + sum_v sigma_v(b + J u), both contacts of a pair see the same
    constraint-space acceleration and answer with the closed-form force of the elliptic cone (cone_qp.primal_force); Newton with an exact line search"""
    from scipy.optimize import brentq
    nc, W, R, b, mus, Alat = P["pairs"], P["W1"], P["R"][:3 * P["pairs"]], P["b"][:3 * P["pairs"]], P["mu"], P["Alat"]
    G = np.zeros((3 * nc, nc))
    for c in range(nc):
        w_, v_ = np.linalg.eigh(Alat[3 * c:3 * c + 3, 3 * c:3 * c + 3]); G[3 * c:3 * c + 3, c] = v_[:, -1] * np.sqrt(max(w_[-1], 0))
    Gp = np.linalg.pinv(G)
    H0 = np.block([[np.linalg.inv(P["Li"]), np.zeros((6, nc))], [np.zeros((nc, 6)), np.linalg.inv(Gp @ Alat @ Gp.T)]])
    J = np.hstack([W, G]); u = np.zeros(6 + nc); hist = []
    phi = lambda y, c, v: primal_force(y[3 * c:3 * c + 3], R[3 * c:3 * c + 3], mus[v])[0]                                         # noqa: E731

    def forces(u):
        y = b + J @ u
        return y, np.concatenate([phi(y, c, c) for c in range(nc)]), np.concatenate([phi(y, c, nc + c) for c in range(nc)])

    def grad(u):
        _, fa, fb = forces(u)
        return H0 @ u - J.T @ (fa + fb)
    for _ in range(iters):
        y, fa, fb = forces(u); T = np.zeros((3 * nc, 3 * nc))
        for c in range(nc):
            for k in range(3):
                e = np.zeros(3 * nc); e[3 * c + k] = 1e-6
                T[3 * c:3 * c + 3, 3 * c + k] = -sum(phi(y + e, c, v) - phi(y - e, c, v) for v in (c, nc + c)) / 2e-6
        du = -np.linalg.solve(H0 + J.T @ (.5 * (T + T.T)) @ J, H0 @ u - J.T @ (fa + fb))
        slope = lambda t: grad(u + t * du) @ du                                                                                    # noqa: E731
        u = u + (1.0 if slope(1.0) <= 0 else brentq(slope, 0.0, 1.0)) * du
        _, fa, fb = forces(u); hist.append(np.concatenate([fa, fb]))
    return hist


def errs(sets, fn, cut=None):
    out = []
    for probs, ex in sets.values():
        e = np.array([np.abs(net_force(p, fn(p)) - net_force(p, x)).max() for p, x in list(zip(probs, ex))[:cut]])
        out.append(f"{np.median(e):.0e}/{np.quantile(e, .99):.0e}/{e.max():.0e}")
    return "   ".join(out)


def by_iteration(sets, fn, its, cut=None):
    probs, ex = list(sets.values())[-1]
    E = np.array([[np.abs(net_force(p, h) - net_force(p, x)).max() for h in fn(p)] for p, x in list(zip(probs, ex))[:cut]])
    return "  ".join(f"{k}: {np.median(E[:, k - 1]):.0e}/{np.quantile(E[:, k - 1], .99):.0e}/{E[:, k - 1].max():.0e}" for k in its)


def Kmat(kn, knt, kt=1.0):
    return np.array([[kn, knt, knt], [knt, kt, kt], [knt, kt, kt]])


def without_coupling(P, kind):
    nc = P["pairs"]; Aarm, Alat = P["Aarm"].copy(), P["Alat"].copy(); mask = np.kron(np.eye(nc), np.ones((3, 3)))
    if kind in ("arm", "all"):
        Aarm *= mask
    if kind in ("lattice", "all"):
        Alat *= mask
    A = Aarm + Alat; A2 = np.block([[A, A], [A, A]])
    return {**P, "A": A2, "Q": A2 + np.diag(P["R"]), "Aarm": Aarm, "Alat": Alat}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "build":
        build_sets(); sys.exit(0)
    sets = load_sets()
    print("explicit pairs; net-force error against the exact optimum (N) as median / 99 % / worst; problem sets: " + " | ".join(sets), flush=True)
    if what in ("t", "all"):
        tr = []
        for probs, _ in sets.values():
            for p in probs:
                jacobi(p, 24, trace=tr)
        tr = np.array(tr)
        print("step length t of round 5's line search by iteration (median, 10 %, 90 % over all problems):  " +
              "  ".join(f"{it}: {np.median(x):.2f} {np.quantile(x, .1):.2f} {np.quantile(x, .9):.2f}" for it in (0, 1, 2, 4, 8, 16, 23) for x in [tr[tr[:, 0] == it, 1]]))
    if what in ("coupling", "all"):
        probs = [p for p in list(sets.values())[-1][0] if p["pairs"] >= 2][:150]
        for kind in ("none", "arm", "lattice", "all"):
            rj, rp = [], []
            for p in probs:
                P2 = without_coupling(p, kind); x = solve_exact(P2)
                e8, e16 = np.abs(jacobi(P2, 8) - x).max(), np.abs(jacobi(P2, 16) - x).max()
                h = pair_block(P2, 10, 2, history=True); e4, e10 = np.abs(h[3] - x).max(), np.abs(h[9] - x).max()
                if e8 > 1e-9:
                    rj.append((e16 / e8) ** (1 / 8))
                if e4 > 1e-9:
                    rp.append((e10 / e4) ** (1 / 6))
            print(f"contraction per pass, coupling between pairs removed: {kind:8s} round 5's Jacobi median {np.median(rj):.2f} 90 % {np.quantile(rj, .9):.2f} | "
                  f"pair blocks median {np.median(rp):.2f} 90 % {np.quantile(rp, .9):.2f} max {np.max(rp):.2f}", flush=True)
    if what in ("cands", "all"):
        cands = [(f"jacobi (round 5), {k}", lambda p, k=k: jacobi(p, k)) for k in (8, 10, 12, 16, 24)]
        cands += [(f"model kn={kn} knt={knt}, {k}", lambda p, k=k, kn=kn, knt=knt: jacobi(p, k, K=Kmat(kn, knt))) for kn, knt in ((2, 1), (2, 1.5), (1.5, 1)) for k in (12, 16)]
        cands += [(f"arm part x (1 + {s} (pairs - 1)), {k}", lambda p, k=k, s=s: jacobi(p, k, arm_scale=s)) for s in (0.3, 1.0) for k in (12, 16)]
        cands += [(f"extrapolated by {bt}, {k}", lambda p, k=k, bt=bt: jacobi(p, k, beta=bt)) for bt in (0.2, 0.4) for k in (12, 16)]
        cands += [(f"gs inside the pair, {k}", lambda p, k=k: pair_block(p, k, 1)) for k in (8, 10, 12)]
        cands += [(f"cheap visit: friction step clamped, no multiplier, {k}", lambda p, k=k: jacobi(p, k, visit=local_solve_cheap)) for k in (16, 24)]
        cands += [(f"cheap visit: no first ray, {k}", lambda p, k=k: jacobi(p, k, visit=lambda *a: local_solve_cheap(*a, ray1=False))) for k in (16, 24)]
        for name, fn in cands:
            print(f"  {name:44s} " + errs(sets, fn), flush=True)
    if what in ("pair", "all"):
        print("last set by pass, exact pair blocks (40 inner rounds):  " + by_iteration(sets, lambda p: pair_block(p, 12, 40, history=True), (4, 6, 8, 10, 12), 250))
        print("last set by pass, two inner rounds (four visits):       " + by_iteration(sets, lambda p: pair_block(p, 12, 2, history=True), (4, 6, 8, 10, 12), 250))
        print("last set by pass, round 5's Jacobi:                     " + by_iteration(sets, lambda p: jacobi(p, 24, history=True), (8, 12, 16, 20, 24), 250))
    if what in ("newton", "all"):
        print("last set by iteration, primal Newton + exact line search: " + by_iteration(sets, lambda p: primal_newton(p, 7), (2, 3, 4, 5, 6, 7), 120))
