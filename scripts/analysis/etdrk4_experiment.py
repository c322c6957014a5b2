"""CPU experiment for DESIGN.md section 8: does an exponential integrator for the stiff oxygen mode (ETD-RK4 of Cox & Matthews on
So, classical RK4 on the other 13 components, same four stage evaluations) allow fewer substeps per control interval than
RK4 x 10 at the 1e-5 gate?  Test infrastructure only (uses oracle/); nothing in the product depends on it."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sbr_oracle as O                     # noqa: E402
from tests.conftest import EPISODES, golden            # noqa: E402

lib, p = O.lib(), O.default_params()
lib.sbro_rhs_reaction.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
lib.sbro_rk4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_double, C.c_double,
                         C.POINTER(C.c_double)]
scale = np.array([1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10.])
SO = 8


def gate(x, ref):
    return (np.abs(x - ref) / (1e-5 * np.abs(ref) + 1e-5 * scale)).max()


def f(x, kla, ec):
    d = np.empty(14)
    lib.sbro_rhs_reaction(C.byref(p), O._p(np.ascontiguousarray(x)), kla, ec, O._p(d))
    return d


def so_rate(x, kla, ec):
    """-d(dSo/dt)/dSo by central differences (the product would use the analytic Monod slopes)."""
    d = 1e-6
    xp, xm = x.copy(), x.copy()
    xp[SO] += d; xm[SO] -= d
    return -(f(xp, kla, ec)[SO] - f(xm, kla, ec)[SO]) / (2 * d)


def phi_coeffs(z):
    """ETD-RK4 weights for one component with linear rate c, z = c*h (c <= 0); Taylor limits for small |z|."""
    if abs(z) < 1e-2:                       # series (Kassam & Trefethen): the closed forms cancel catastrophically near 0
        e2 = np.exp(z / 2)
        q = 0.5 * (1 + z / 4 + z * z / 24)                                  # (e^{z/2}-1)/z
        return (e2, np.exp(z), q, 1 / 6 + z / 6 + 3 * z * z / 40, 1 / 6 + z / 12 + z * z / 40, 1 / 6 - z * z / 120)
    e2, e1 = np.exp(z / 2), np.exp(z)
    q = (e2 - 1) / z
    f1 = (-4 - z + e1 * (4 - 3 * z + z * z)) / z ** 3
    f2 = (2 + z + e1 * (-2 + z)) / z ** 3
    f3 = (-4 - 3 * z - z * z + e1 * (4 - z)) / z ** 3
    return e2, e1, q, f1, f2, f3


def etdrk4_step(x, h, kla, ec, refresh_L=True, L=None):
    """One step: component SO uses u' = -L u + N(x), N = f_So + L*So with L frozen at the start of the step."""
    if L is None or refresh_L:
        L = max(so_rate(x, kla, ec), 0.0)
    z = -L * h
    e2, e1, q, f1, f2, f3 = phi_coeffs(z) if L > 0 else (1, 1, 0.5, 1 / 6, 1 / 3, 1 / 6)

    def N(y):
        d = f(y, kla, ec)
        d[SO] = d[SO] + L * y[SO]
        return d
    n1 = N(x)
    a = x + 0.5 * h * n1
    a[SO] = x[SO] * e2 + h * q * n1[SO]
    n2 = N(a)
    b = x + 0.5 * h * n2
    b[SO] = x[SO] * e2 + h * q * n2[SO]
    n3 = N(b)
    c = x + h * n3
    c[SO] = a[SO] * e2 + h * q * (2 * n3[SO] - n1[SO])
    n4 = N(c)
    out = x + h / 6 * (n1 + 2 * n2 + 2 * n3 + n4)
    out[SO] = x[SO] * e1 + h * (f1 * n1[SO] + 2 * f2 * (n2[SO] + n3[SO]) + f3 * n4[SO])
    return out, L


def main():
    ivs = []
    for name in EPISODES:
        e = golden("sbros_" + name)
        for i in range(len(e["iv_kind"])):
            ivs.append((e["iv_x_start"][i], float(e["iv_t_end"][i]) - float(e["iv_t_start"][i]), float(e["iv_Kla"][i]),
                        float(e["iv_EC"][i])))
    exact, hard = [], []
    for x0, span, kla, ec in ivs:
        x = x0.copy(); lib.sbro_rk4(C.byref(p), 0, O._p(x), span, 160, kla, ec, None); exact.append(x)
        y = x0.copy(); lib.sbro_rk4(C.byref(p), 0, O._p(y), span, 5, kla, ec, None); hard.append(gate(y, x))
    # the 120 hardest intervals for RK4 plus 120 of the stiffest (anoxic, So ~ 0) ones
    order = list(np.argsort(hard)[::-1][:120])
    stiff = [k for k, (x0, span, kla, ec) in enumerate(ivs) if kla == 0.0 and x0[SO] < 1e-3][::12][:120]
    sel = sorted(set(order + stiff))
    print("%d intervals selected (%d hardest for RK4, %d stiff anoxic)" % (len(sel), len(order), len(stiff)))
    for n in (1, 2, 3, 4, 5):
        for refresh in (True, False):
            worst, wk = 0.0, -1
            for k in sel:
                x0, span, kla, ec = ivs[k]
                x, L = x0.copy(), None
                for s in range(n):
                    x, L = etdrk4_step(x, span / n, kla, ec, refresh_L=refresh or s == 0, L=L)
                g = gate(x, exact[k])
                if not np.isfinite(g): g = 1e30
                if g > worst: worst, wk = g, k
            print("ETD-RK4(So) n=%d (%2d RHS + %d rate evals)  L %s: worst gate %.4f  (interval %d: Kla %.0f, So0 %.3g)"
                  % (n, 4 * n, n if refresh else 1, "per substep " if refresh else "per interval", worst, wk, ivs[wk][2], ivs[wk][0][SO]), flush=True)


if __name__ == "__main__":
    main()
