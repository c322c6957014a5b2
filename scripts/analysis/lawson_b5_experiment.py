"""CPU experiment: integrating-factor (Lawson) version of Butcher's fifth-order scheme - the oxygen
equation is written So' = -L So + N(x) with L frozen per substep, and the scheme is applied to v = exp(L t) So; all other
components are integrated as they are.  Test infrastructure only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from etdrk4_experiment import C, O, EPISODES, golden, lib, p, gate, f, so_rate, SO      # noqa: E402

A = [[], [1 / 4], [1 / 8, 1 / 8], [0, -1 / 2, 1], [3 / 16, 0, 0, 9 / 16], [-3 / 7, 2 / 7, 12 / 7, -12 / 7, 8 / 7]]
B = [7 / 90, 0, 32 / 90, 12 / 90, 32 / 90, 7 / 90]
CN = [sum(r) for r in A]


def lawson_step(x, h, kla, ec, use_L=True):
    L = max(so_rate(x, kla, ec), 0.0) if use_L else 0.0
    v0 = x[SO]                                    # v = exp(L (t - t_n)) So, v(t_n) = So
    ks = []                                       # derivatives of the TRANSFORMED vector (component SO holds v')
    for i in range(6):
        y = x.copy()
        v = v0
        for j in range(i):
            if A[i][j] != 0.0:
                y = y + h * A[i][j] * ks[j]
                v = v + h * A[i][j] * ks[j][SO]
        y[SO] = np.exp(-L * CN[i] * h) * v
        d = f(y, kla, ec)
        d[SO] = np.exp(L * CN[i] * h) * (d[SO] + L * y[SO])
        ks.append(d)
    out, v = x.copy(), v0
    for i in range(6):
        out = out + h * B[i] * ks[i]
        v = v + h * B[i] * ks[i][SO]
    out[SO] = np.exp(-L * h) * v
    return out


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
    order = list(np.argsort(hard)[::-1][:120])
    stiff = [k for k, (x0, span, kla, ec) in enumerate(ivs) if kla == 0.0 and x0[SO] < 1e-3][::12][:120]
    sel = sorted(set(order + stiff))
    for use_L in (False, True):
        for n in (2, 3, 4, 5):
            worst, wk, worst_stiff = 0.0, -1, 0.0
            for k in sel:
                x0, span, kla, ec = ivs[k]
                x = x0.copy()
                for s in range(n):
                    x = lawson_step(x, span / n, kla, ec, use_L)
                g = gate(x, exact[k])
                g = g if np.isfinite(g) else 1e30
                if g > worst: worst, wk = g, k
                if k in stiff: worst_stiff = max(worst_stiff, g)
            print("%s Butcher5 n=%d (%2d RHS): worst gate %.4f (interval %d: Kla %.0f, So0 %.3g) | worst on the stiff anoxic ones %.4f"
                  % ("Lawson" if use_L else "plain ", n, 6 * n, worst, wk, ivs[wk][2], ivs[wk][0][SO], worst_stiff), flush=True)


if __name__ == "__main__":
    main()
