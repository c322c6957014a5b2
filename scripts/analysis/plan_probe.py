"""Round 5: a broad synthetic probe of scheme 1's plan - random plant states far beyond what the fixtures hold (sludge 0.4 - 3 x
the reference's, substrate from exhausted to shock-loaded, oxygen from 1e-12 to saturation, every Kla and EC) - one interval each,
cfg.scheme = 1 against RK4 x 320 (C oracle).  Prints the distribution of the gate and the worst cases with their plans.

    python scripts/analysis/plan_probe.py [n] [seed]

Test infrastructure / analysis only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import gate  # noqa: E402
from oracle import sbr_oracle as O  # noqa: E402
from oracle import sbr_params as P  # noqa: E402


def sample(rs):
    x = np.empty(14)
    x[0] = rs.uniform(0.6, 1.4)
    x[1] = rs.uniform(10, 40)
    x[2] = 10 ** rs.uniform(-1.5, 1.8)            # Ss 0.03 .. 63
    x[3] = rs.uniform(300, 1600)
    x[4] = 10 ** rs.uniform(0.7, 2.6)             # Xs 5 .. 400
    x[5] = rs.uniform(500, 4000)                  # Xbh
    x[6] = rs.uniform(30, 300)                    # Xba
    x[7] = rs.uniform(100, 700)
    x[8] = rs.choice([0.0, 1e-12, 1e-7, 1e-4, 10 ** rs.uniform(-3, 0.9)], p=[0.1, 0.1, 0.1, 0.1, 0.6])
    x[9] = 10 ** rs.uniform(-3, 1.5)              # Sno
    x[10] = 10 ** rs.uniform(-2, 1.6)             # Snh 0.01 .. 40
    x[11] = rs.uniform(0.05, 8)
    x[12] = rs.uniform(0.5, 12)
    x[13] = rs.uniform(2, 9)
    kla = rs.choice([0.0, 0.0, rs.uniform(0, 5), rs.uniform(5, 240), 240.0])
    ec = rs.choice([0.0, 0.0, rs.uniform(0, 5e-4), 5e-4]) if kla == 0 else 0.0
    return x, float(kla), float(ec)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    span = (0.25 + P.T_DELTA) - 0.25
    p1 = O.default_params(scheme=1)
    g, g0, plans, states = np.empty(n), np.empty(n), np.empty(n, dtype=int), []
    for i in range(n):
        x, kla, ec = sample(rs)
        x1, nn = O.reaction_interval(x, span, kla, ec, params=p1, scheme=1)
        ex = O.rk4(0, x, span, 320, kla, ec)
        ok = np.isfinite(ex).all() and (ex[[2, 4, 5, 8, 9, 10]] > -1e-9).all()         # the fine solution itself stays in the domain
        g[i] = gate(x1, ex).max() if ok else np.nan
        g0[i] = gate(O.rk4(0, x, span, 10, kla, ec), ex).max() if ok else np.nan
        plans[i] = nn
        states.append((x, kla, ec))
    v = np.isfinite(g)
    print("%d random intervals (%d with an in-domain fine solution): worst %.3g, p99.9 %.3g, p99 %.3g, median %.2e; > 1: %d, > 0.5: %d"
          % (n, v.sum(), np.nanmax(g), np.nanpercentile(g, 99.9), np.nanpercentile(g, 99), np.nanmedian(g), (g > 1).sum(), (g > 0.5).sum()))
    print("RK4 x 10 on the same intervals: worst %.3g, p99.9 %.3g, p99 %.3g, median %.2e; > 1: %d, > 0.5: %d"
          % (np.nanmax(g0), np.nanpercentile(g0, 99.9), np.nanpercentile(g0, 99), np.nanmedian(g0), (g0 > 1).sum(), (g0 > 0.5).sum()))
    print("plans:", {int(k): int((plans[v] == k).sum()) for k in np.unique(plans[v])})
    for i in np.argsort(np.where(v, g, -1))[-8:][::-1]:
        x, kla, ec = states[i]
        print("  gate %.3g (RK4 x 10: %.3g)  n=%d  So %.3g Ss %.3g Snh %.3g Sno %.3g Xbh %.0f Xba %.0f Xs %.0f Kla %.3g EC %.2g" % (
            g[i], g0[i], plans[i], x[8], x[2], x[10], x[9], x[5], x[6], x[4], kla, ec))


if __name__ == "__main__":
    main()
