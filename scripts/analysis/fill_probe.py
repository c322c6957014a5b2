"""Round 5, late: can the FILL intervals be integrated by the adaptive Butcher-5 scheme after all, now that the plan projects
Ss and Snh along their start slopes (plan v2: the inflow is part of those slopes)?  The first attempt (plan v1) was taken back
when a carried-over SBR-v2 cycle showed fill intervals 70 gates off.  Populations: the fill intervals of SBR-v2 cycles (fresh and
carried over, scripts/analysis/cycle_intervals.py) and the 26 macro intervals of SBROS-v1 resets on all eight scenarios, each
judged on its own against RK4 x 160.  Candidates: plan v2 as it is; with a floor of 2 / 4 steps.
    python scripts/analysis/fill_probe.py        (test infrastructure / analysis only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402
from oracle import sbr_params as P  # noqa: E402
from oracle import sbr_ref as R  # noqa: E402

SCALE = np.array([1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10])


def gate(a, b):
    return np.max(np.abs(a - b) / (1e-5 * np.abs(b) + 1e-5 * SCALE))


def rk4(f, x, span, n):
    h = span / n
    for _ in range(n):
        k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
        x = x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def b5a_fill(x, span, kla, loading, floor):
    f = lambda y: R.rhs_fill(y, 0.0, kla, loading)          # noqa: E731
    k1 = f(x)
    n, slaved, lam0 = R.b5a_plan(x, k1, span, kla)
    slaved = False                                           # the inflow carries oxygen: never held
    n = max(n, floor)
    h = span / n
    for s in range(n):
        if s > 0:
            k1 = f(x)
        x = R.b5_step(f, x, h, k1, False)
    return x, n


def population():
    items = []
    path = "/tmp/sbr_cycle_intervals.npz"
    if os.path.exists(path):
        d = np.load(path)
        for j in np.nonzero(d["kind"] == 1)[0]:
            items.append(("cycle", d["X"][:, j].copy(), float(d["span"][j]), float(d["kla"][j]), d["loading"][:, j].copy()))
    t = golden("influent_tables")
    rs = np.random.RandomState(3)
    for scen in range(8):
        for rep in range(2):
            infl = R.influent_mix(t["means"][scen], t["stds"][scen], rs.randn(48))
            x = np.array(P.X0_INIT, dtype=np.float64)
            qin = P.WV - x[0]
            loading = infl.copy(); loading[0] = qin / P.T1_END
            span = P.T1_END / 26
            for m in range(26):
                items.append(("reset s%d" % scen, x.copy(), span, 0.0, loading))
                x = rk4(lambda y: R.rhs_fill(y, 0.0, 0.0, loading), x, span, 40)
    return items


if __name__ == "__main__":
    items = population()
    print(len(items), "fill intervals")
    for floor in (1, 2, 4):
        g, ns, g10 = [], [], []
        for tag, x, span, kla, loading in items:
            f = lambda y: R.rhs_fill(y, 0.0, kla, loading)  # noqa: E731
            ref = rk4(f, x, span, 160)
            y, n = b5a_fill(x.copy(), span, kla, loading, floor)
            g.append(gate(y, ref)); ns.append(n)
            if floor == 1:
                g10.append(gate(rk4(f, x, span, 10), ref))
        g, ns = np.array(g), np.array(ns)
        print("floor %d: worst %.3f  p99 %.3f  >0.1: %d  mean steps %.2f  (counts %s)" % (
            floor, g.max(), np.percentile(g, 99), (g > 0.1).sum(), ns.mean(), dict(zip(*np.unique(ns, return_counts=True)))))
        if floor == 1:
            g10 = np.array(g10)
            print("RK4 x 10: worst %.4f  p99 %.4f" % (g10.max(), np.percentile(g10, 99)))
            w = int(np.argmax(g))
            print("  worst case:", items[w][0], "n", ns[w], "So %.3g Ss %.3g Snh %.3g V %.3g kla %.3g" % (items[w][1][8], items[w][1][2], items[w][1][10], items[w][1][0], items[w][3]))
