"""Round 5, VERDICT item 3: a bounded, oracle-side study of integrating one control interval with fewer right-hand-side
evaluations than classical RK4 x 10 (40).  Test infrastructure: uses oracle/ and the committed fixtures only, numpy-vectorised
over every reference-captured interval (the 2 796 golden ones + the scenario episodes' valid ones, ~10 900).

    python scripts/analysis/rhs_study.py survey      error of plain schemes per interval, by regime
    python scripts/analysis/rhs_study.py ...         see main()

The library of schemes (Butcher tableaux) and the vectorised right-hand side live here so that later experiments import them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EPISODES, SCENARIO_EPISODES, golden, valid_calls  # noqa: E402
from oracle import sbr_params as P  # noqa: E402
from oracle import sbr_ref as R  # noqa: E402

SCALE = np.asarray(P.STATE_SCALE, dtype=np.float64)
COMP = "V Si Ss Xi Xs Xbh Xba Xp So Sno Snh Snd Xnd Salk".split()
DT = P.DT


def gate(x, ref):
    """[14, N] -> [N]: worst component of the parity gate."""
    return (np.abs(x - ref) / (1e-5 * np.abs(ref) + 1e-5 * SCALE[:, None])).max(axis=0)


def gate_comp(x, ref):
    return np.abs(x - ref) / (1e-5 * np.abs(ref) + 1e-5 * SCALE[:, None])


def f(x, kla, ec):
    """reaction_dxdt (gym_SBR_oneshot.py:1658-1787) on x [14, N]: oracle/sbr_ref.py's conversion() is written component-wise
    and broadcasts."""
    r = R.conversion(x, kla)
    d = np.empty_like(x)
    d[0] = ec
    q = ec / x[0]
    for i in range(1, 14):
        d[i] = r[i] + q * ((P.EC_CONC - x[i]) if i == 2 else -x[i])
    return d


def load_intervals(names=None, with_names=False):
    X, span, kla, ec, kind, ref, tag, call = [], [], [], [], [], [], [], []
    for name in (names or EPISODES + SCENARIO_EPISODES):
        e = golden("sbros_" + name)
        nv = valid_calls(e)
        keep = e["iv_call"] <= nv
        X.append(e["iv_x_start"][keep]); ref.append(e["iv_x_end"][keep])
        span.append((e["iv_t_end"] - e["iv_t_start"])[keep]); kla.append(e["iv_Kla"][keep]); ec.append(e["iv_EC"][keep])
        kind.append(e["iv_kind"][keep]); call.append(e["iv_call"][keep]); tag += [name] * int(keep.sum())
    out = dict(X=np.concatenate(X).T.copy(), span=np.concatenate(span), kla=np.concatenate(kla), ec=np.concatenate(ec),
               kind=np.concatenate(kind), ref=np.concatenate(ref).T.copy(), call=np.concatenate(call), tag=np.asarray(tag))
    return out


def erk_step(A, b, x, h, kla, ec, rhs=f):
    ks = []
    for i in range(len(b)):
        y = x
        for j in range(i):
            if A[i][j] != 0.0:
                y = y + (h * A[i][j]) * ks[j]
        ks.append(rhs(y, kla, ec))
    out = x
    for i in range(len(b)):
        if b[i] != 0.0:
            out = out + (h * b[i]) * ks[i]
    return out


def integrate(scheme, x, span, n, kla, ec, rhs=f):
    A, b = scheme
    h = span / n
    for _ in range(n):
        x = erk_step(A, b, x, h, kla, ec, rhs)
    return x


RK4 = ([[], [.5], [0, .5], [0, 0, 1]], [1 / 6, 1 / 3, 1 / 3, 1 / 6])
B5 = ([[], [1 / 4], [1 / 8, 1 / 8], [0, -1 / 2, 1], [3 / 16, 0, 0, 9 / 16], [-3 / 7, 2 / 7, 12 / 7, -12 / 7, 8 / 7]],
      [7 / 90, 0, 32 / 90, 12 / 90, 32 / 90, 7 / 90])
s21 = np.sqrt(21)
B6 = ([[], [1], [3 / 8, 1 / 8], [8 / 27, 2 / 27, 8 / 27], [3 * (3 * s21 - 7) / 392, -8 * (7 - s21) / 392, 48 * (7 - s21) / 392, -3 * (21 - s21) / 392],
       [-5 * (231 + 51 * s21) / 1960, -40 * (7 + s21) / 1960, -320 * s21 / 1960, 3 * (21 + 121 * s21) / 1960, 392 * (6 + s21) / 1960],
       [15 * (22 + 7 * s21) / 180, 120 / 180, 40 * (7 * s21 - 5) / 180, -63 * (3 * s21 - 2) / 180, -14 * (49 + 9 * s21) / 180, 70 * (7 - s21) / 180]],
      [9 / 180, 0, 64 / 180, 0, 49 / 180, 49 / 180, 9 / 180])
SCHEMES = {"RK4": RK4, "B5": B5, "B6": B6}


def exact(iv, n=160):
    return integrate(RK4, iv["X"], iv["span"], n, iv["kla"], iv["ec"])


def so_rate(x, kla):
    """-d(dSo/dt)/dSo, analytic: the relaxation rate of dissolved oxygen (the stiff mode)."""
    ss, xbh, xba, so, snh = x[2], x[5], x[6], x[8], x[10]
    a1 = (1 - P.YH) / P.YH * P.MUH * ss / (P.KS + ss) * xbh
    a3 = (4.57 - P.YA) / P.YA * P.MUA * snh / (P.KNH + snh) * xba
    return a1 * P.KOH / (P.KOH + so) ** 2 + a3 * P.KOA / (P.KOA + so) ** 2 + kla


def survey():
    iv = load_intervals()
    n_iv = len(iv["span"])
    ex = exact(iv)
    print("%d intervals; reference (LSODA default) vs RK4-160: worst %.3f" % (n_iv, gate(iv["ref"], ex).max()))
    lam = so_rate(iv["X"], iv["kla"])
    print("oxygen relaxation rate at interval starts: max %.0f /d (x dt = %.2f); anoxic max %.0f, aerobic max %.0f" % (
        lam.max(), lam.max() * DT, lam[iv["kind"] == 0].max(), lam[iv["kind"] == 1].max()))
    for name, n_list in (("RK4", (3, 4, 5, 6, 8, 10)), ("B5", (1, 2, 3, 4, 5)), ("B6", (1, 2, 3, 4))):
        for n in n_list:
            with np.errstate(all="ignore"):
                x = integrate(SCHEMES[name], iv["X"], iv["span"], n, iv["kla"], iv["ec"])
            g = gate(x, ex)
            g = np.where(np.isfinite(g), g, 1e30)
            stages = len(SCHEMES[name][1])
            an, ae = iv["kind"] == 0, iv["kind"] == 1
            print("%-4s n=%2d (%2d RHS): worst %9.3g  share <= 0.5: %.4f (anoxic %.4f, aerobic %.4f)  p50 %.2e p99 %.2e"
                  % (name, n, stages * n, g.max(), (g <= 0.5).mean(), (g[an] <= 0.5).mean(), (g[ae] <= 0.5).mean(),
                     np.median(g), np.percentile(g, 99)), flush=True)



# ------------------------------------------------------------------------------------------------ the adaptive scheme
SO_SLAVED, Z1, Z2, Z_STAB = 1e-6, 0.3, 1.0, 3.0


def oxygen_rate_parts(x):
    """a1, a3 of  -d(dSo/dt)/dSo = a1 Koh/(Koh+So)^2 + a3 Koa/(Koa+So)^2 + Kla  (from reaction_dxdt's rho1 and rho3)."""
    ss, xbh, xba, snh = x[2], x[5], x[6], x[10]
    a1 = (1 - P.YH) / P.YH * P.MUH * (ss / (P.KS + ss)) * xbh
    a3 = (4.57 - P.YA) / P.YA * P.MUA * (snh / (P.KNH + snh)) * xba
    return a1, a3


def lam_of(a1, a3, kla, so):
    return a1 * P.KOH / ((P.KOH + so) * (P.KOH + so)) + a3 * P.KOA / ((P.KOA + so) * (P.KOA + so)) + kla


def f_w(y, v0, kla, ec):
    """reaction_dxdt in scaled-mass variables (oracle/sbr_ref.py rhs_reaction_w), vectorised."""
    s = y[0] / v0
    c = y / s
    c[0] = y[0]
    r = R.conversion(c, kla)
    d = np.empty_like(y)
    d[0] = ec
    q0 = ec / v0
    for i in range(1, 14):
        d[i] = s * r[i] + (q0 * P.EC_CONC if i == 2 else 0.0)
    return d


def b5a_plan(x, span, kla, ec):
    """Per interval: (n, slaved, fallback, lam0, k1) of the adaptive scheme."""
    v0 = x[0]
    k1 = np.where(ec != 0, f_w(x, v0, kla, ec), f(x, kla, ec))
    a1, a3 = oxygen_rate_parts(x)
    so = x[8]
    slaved = (kla == 0) & (np.abs(so) < SO_SLAVED)
    so_lo = np.maximum(0.0, np.minimum(so, so + k1[8] * span))
    z_ub = lam_of(a1, a3, kla, so_lo) * span
    lam0 = lam_of(a1, a3, kla, 0.0)
    n = np.where(slaved | (z_ub < Z1), 1, np.where(z_ub < Z2, 2, 4))
    fallback = ~slaved & (n == 4) & (lam0 * span / 4 > Z_STAB)
    return n, slaved, fallback, lam0, z_ub


def b5a_interval(x, span, kla, ec):
    """One control interval by the adaptive scheme, vectorised over intervals (grouped by step count)."""
    n, slaved, fallback, lam0, _ = b5a_plan(x, span, kla, ec)
    out = np.empty_like(x)
    A, b = B5
    for nn in (1, 2, 4):
        for dose in (False, True):
            m = (n == nn) & ((ec != 0) == dose) & ~fallback
            if not m.any():
                continue
            xs, sp, kl, e_, sl = x[:, m], span[m], kla[m], ec[m], slaved[m]
            v0 = xs[0].copy()
            mask = np.ones_like(xs)
            mask[8, sl] = 0.0

            def rhs(y, kla_, ec_, v0=v0, mask=mask, dose=dose):
                return (f_w(y, v0, kla_, ec_) if dose else f(y, kla_, ec_)) * mask
            y = integrate(B5, xs, sp, nn, kl, e_, rhs=rhs)
            if dose:
                s_end = y[0] / v0
                y[1:] = y[1:] / s_end
            y[8, sl] = y[8, sl] / (1.0 + lam0[m][sl] * sp[sl])
            out[:, m] = y
    if fallback.any():
        m = fallback
        out[:, m] = integrate(RK4, x[:, m], span[m], 10, kla[m], ec[m])
    return out, n, slaved, fallback


def adaptive():
    iv = load_intervals()
    ex = exact(iv)
    x, n, slaved, fb = b5a_interval(iv["X"], iv["span"], iv["kla"], iv["ec"])
    g, gr = gate_comp(x, ex), gate_comp(x, iv["ref"])
    w = g.max(axis=0)
    j = int(w.argmax())
    print("adaptive scheme, %d intervals: slaved %d, n=1 %d, n=2 %d, n=4 %d, fallback %d; mean RHS evaluations per interval %.2f"
          % (len(w), slaved.sum(), ((n == 1) & ~slaved).sum(), (n == 2).sum(), (n == 4).sum(), fb.sum(), 6.0 * n.mean()))
    print("  vs RK4-160: worst %.4f (%s, %s call %d, n=%d) p99 %.2e p50 %.2e" % (w.max(), COMP[int(g[:, j].argmax())], iv["tag"][j],
                                                                               iv["call"][j], n[j], np.percentile(w, 99), np.median(w)))
    print("  vs the reference's LSODA end state: worst %.4f" % gr.max())
    rk = integrate(RK4, iv["X"], iv["span"], 10, iv["kla"], iv["ec"])
    print("  RK4 x 10 on the same intervals: worst vs RK4-160 %.4f, vs reference %.4f" % (gate(rk, ex).max(), gate(rk, iv["ref"]).max()))
    for nn, sl in ((1, True), (1, False), (2, False), (4, False)):
        m = (n == nn) & (slaved == sl)
        print("    n=%d%s: %5d intervals, worst %.4f" % (nn, " slaved" if sl else "", m.sum(), w[m].max()))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "survey"
    {"survey": survey, "adaptive": adaptive}[cmd]()
