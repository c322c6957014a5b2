"""Round 5: design of a 6-stage, fifth-order explicit Runge-Kutta scheme whose stability polynomial
R(z) = 1 + z + ... + z^5/120 + (m/720) z^6 has a LONG real stability interval (m = 1.125 is Butcher's scheme: 3.39; m ~ 0.6: 5.1)
and a small principal error.  Rooted trees are generated to order 6, the 17 order conditions to order 5 and the coefficient of
z^6 are imposed as equality constraints, and the 2-norm of the order-6 residuals (the principal error) is minimised.

    python scripts/analysis/rk_design.py [m]

Test infrastructure / analysis only.
"""
import itertools
import math
import sys

import numpy as np
from scipy.optimize import minimize


def trees_of_order(n, _cache={}):
    """All rooted trees with n vertices as canonical nested tuples (a tree = sorted tuple of its root's subtrees)."""
    if n in _cache:
        return _cache[n]
    if n == 1:
        out = [()]
    else:
        out = set()
        # partitions of n-1 into subtree sizes
        def parts(total, maxpart):
            if total == 0:
                yield ()
                return
            for p in range(min(total, maxpart), 0, -1):
                for rest in parts(total - p, p):
                    yield (p,) + rest
        for part in parts(n - 1, n - 1):
            pools = [trees_of_order(p) for p in part]
            for combo in itertools.product(*pools):
                out.add(tuple(sorted(combo)))
        out = sorted(out)
    _cache[n] = out
    return out


def order(t):
    return 1 + sum(order(u) for u in t)


def gamma(t):
    g = order(t)
    for u in t:
        g *= gamma(u)
    return g


def phi(t, A):
    """Stage vector of the elementary weight: phi(leaf) = e, phi([t1..tm])_i = prod_k (A phi(t_k))_i."""
    s = A.shape[0]
    v = np.ones(s)
    for u in t:
        v = v * (A @ phi(u, A))
    return v


def unpack(p, s=6):
    A = np.zeros((s, s))
    k = 0
    for i in range(1, s):
        for j in range(i):
            A[i, j] = p[k]
            k += 1
    b = p[k:k + s]
    return A, np.asarray(b)


def pack(A, b):
    s = len(b)
    return np.array([A[i][j] for i in range(1, s) for j in range(i)] + list(b), dtype=float)


TALL6 = ((((((),),),),),)


def residuals(p, upto, m):
    A, b = unpack(p)
    out = []
    for n in range(1, upto + 1):
        for t in trees_of_order(n):
            target = 1.0 / gamma(t)
            if t == TALL6:
                target = m / 720.0
            out.append(b @ phi(t, A) - target)
    return np.array(out)


def principal_error(p, m):
    r = residuals(p, 6, m)
    n5 = sum(len(trees_of_order(n)) for n in range(1, 6))
    return r[n5:]


def stability_limit(A, b):
    s = len(b)
    coeffs, v = [1.0], np.ones(s)
    for k in range(1, s + 1):
        coeffs.append(b @ v)
        v = A @ v
    z = -np.linspace(0, 12, 240001)
    R = sum(c * z ** k for k, c in enumerate(coeffs))
    bad = np.where(np.abs(R) > 1 + 1e-12)[0]
    return (-z[bad[0]] if len(bad) else 12.0), coeffs


B5_A = [[], [1 / 4], [1 / 8, 1 / 8], [0, -1 / 2, 1], [3 / 16, 0, 0, 9 / 16], [-3 / 7, 2 / 7, 12 / 7, -12 / 7, 8 / 7]]
B5_b = [7 / 90, 0, 32 / 90, 12 / 90, 32 / 90, 7 / 90]


def full(Arows):
    s = len(Arows)
    A = np.zeros((s, s))
    for i, r in enumerate(Arows):
        A[i, :len(r)] = r
    return A


def design(m, zero_mask=None, starts=40, seed=0, verbose=True):
    """zero_mask: indices of the packed vector forced to 0 (sparsity)."""
    rs = np.random.RandomState(seed)
    n5 = sum(len(trees_of_order(n)) for n in range(1, 6))
    best = None
    p_b5 = pack(full(B5_A), B5_b)
    free = np.ones(21, dtype=bool)
    if zero_mask is not None:
        free[list(zero_mask)] = False

    def expand(q):
        p = np.zeros(21)
        p[free] = q
        return p

    def cons(q):
        p = expand(q)
        r = residuals(p, 5, m)
        A, b = unpack(p)
        tall = b @ phi(TALL6, A) - m / 720.0
        return np.append(r, tall)

    def obj(q):
        p = expand(q)
        e = principal_error(p, m)
        return float(e @ e) + 1e-7 * float(q @ q)

    for k in range(starts):
        q0 = p_b5[free] + (0.0 if k == 0 else 0.3) * rs.randn(free.sum())
        try:
            res = minimize(obj, q0, method="SLSQP", constraints=[{"type": "eq", "fun": cons}],
                           options={"maxiter": 500, "ftol": 1e-16})
        except Exception:
            continue
        c = np.abs(cons(res.x)).max()
        if c < 1e-11:
            p = expand(res.x)
            e = principal_error(p, m)
            score = float(np.sqrt(e @ e))
            if best is None or score < best[0]:
                best = (score, p)
                if verbose:
                    A, b = unpack(p)
                    print("start %d: principal error norm %.3e, max |coef| %.2f, limit %.3f" % (k, score, np.abs(p).max(), stability_limit(A, b)[0]),
                          flush=True)
    return best


if __name__ == "__main__":
    m = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
    A5, b5 = full(B5_A), np.array(B5_b)
    e = principal_error(pack(A5, b5), 1.125)
    print("Butcher5: order-5 residual %.2e, principal error norm %.3e (incl. tall tree %.3e), limit %.3f" % (
        np.abs(residuals(pack(A5, b5), 5, 1.125)).max(), np.sqrt(e @ e), 0.125 / 720, stability_limit(A5, b5)[0]))
    best = design(m)
    if best:
        A, b = unpack(best[1])
        np.set_printoptions(precision=17, linewidth=200)
        print("A =\n", A, "\nb =", b, "\nc =", A.sum(axis=1))
        np.save("/tmp/s5_m%.3f.npy" % m, best[1])
