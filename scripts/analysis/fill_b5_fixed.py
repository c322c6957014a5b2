"""Round 6, VERDICT r5 item 7: price the fill phase of reset() (252 RK4 substeps at h = dt: 1 008 right-hand sides) as FIXED-step
Butcher-5 at h = 2 dt (126 steps: 756) - no plan, unlike round 5's adaptive attempt.  Post-fill state against the reference's own
(LSODA) on the 34 reference episodes and against RK4 x 2520, plus random influents on every scenario; the worst oxygen rate times
the step along the way (Butcher-5 is stable on the real axis to 3.39).      python scripts/analysis/fill_b5_fixed.py
Test infrastructure / analysis only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EPISODES, HELDOUT_EPISODES, SCENARIO_EPISODES, gate, golden  # noqa: E402
from oracle import sbr_params as P  # noqa: E402
from oracle import sbr_ref as R  # noqa: E402


def rk4(f, x, span, n):
    h = span / n
    for _ in range(n):
        k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
        x = x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def b5(f, x, span, n):
    h = span / n
    lam_h = 0.0
    for _ in range(n):
        ss, xbh, xba, so, snh = x[2], x[5], x[6], x[8], x[10]
        a1 = (1 - P.YH) / P.YH * P.MUH * ss / (P.KS + ss) * xbh
        a3 = (4.57 - P.YA) / P.YA * P.MUA * snh / (P.KNH + snh) * xba
        lam_h = max(lam_h, (a1 / P.KOH + a3 / P.KOA) * h)
        x = R.b5_step(f, x, h, f(x), False)
    return x, lam_h


def main():
    t_fill, rows = P.T1_END, 252
    worst = {"rk4_252": 0.0, "b5_126": 0.0, "b5_84": 0.0, "b5_126_vs_fine": 0.0, "rk4_252_vs_fine": 0.0}
    lam = 0.0
    cases = []
    for name in EPISODES + SCENARIO_EPISODES + HELDOUT_EPISODES:
        e = golden("sbros_" + name)
        cases.append((name, e["influent_mixed"].copy(), e["x_postfill"].copy()))
    t = golden("influent_tables")
    rs = np.random.RandomState(11)
    for scen in range(8):
        for rep in range(4):
            infl = R.influent_mix(t["means"][scen], t["stds"][scen], rs.randn(48) * 1.5)
            infl[0] = (P.WV - P.IV_INIT) / t_fill
            cases.append(("rand_s%d_%d" % (scen, rep), infl, None))
    for name, infl, ref in cases:
        x0 = np.array(P.X0_INIT, dtype=np.float64)
        f = lambda y: R.rhs_fill(y, 0.0, 0.0, infl)          # noqa: E731   Kla = 0 during the fill (:1593-1617)
        fine = rk4(f, x0.copy(), t_fill, 2520)
        a = rk4(f, x0.copy(), t_fill, rows)
        b, lh = b5(f, x0.copy(), t_fill, rows // 2)
        c, _ = b5(f, x0.copy(), t_fill, rows // 3)
        lam = max(lam, lh)
        worst["b5_126_vs_fine"] = max(worst["b5_126_vs_fine"], gate(b, fine).max())
        worst["rk4_252_vs_fine"] = max(worst["rk4_252_vs_fine"], gate(a, fine).max())
        if ref is not None:
            worst["rk4_252"] = max(worst["rk4_252"], gate(a, ref).max())
            worst["b5_126"] = max(worst["b5_126"], gate(b, ref).max())
            worst["b5_84"] = max(worst["b5_84"], gate(c, ref).max())
    print("%d fill phases (34 reference episodes + 32 random influents)" % len(cases))
    print("post-fill state vs the reference's (34 episodes): RK4 x 252 worst %.4f; Butcher-5 x 126 (h = 2 dt) worst %.4f; Butcher-5 x 84 (h = 3 dt) worst %.4f"
          % (worst["rk4_252"], worst["b5_126"], worst["b5_84"]))
    print("post-fill state vs RK4 x 2520 (all 66): RK4 x 252 worst %.2e; Butcher-5 x 126 worst %.2e" % (worst["rk4_252_vs_fine"], worst["b5_126_vs_fine"]))
    print("largest lam(0) h along the fill at h = 2 dt: %.2f (Butcher-5 stable to 3.39; RK4 at h = dt: %.2f of 2.785)" % (lam, lam / 2))
    print("right-hand sides: 1008 -> 756 (-25 %%); instructions per fill: 252 x 382 = 96 264 -> 126 x ~700 = ~88 000 (-8 %%): a Butcher-5 step of the\n"
          "filling form costs ~6 x (50 + 18 inflow terms) + 17 x 9 combination FMAs + 5 reciprocals of V at its stage times")


if __name__ == "__main__":
    main()
