"""CPU oracle, layer 1: NumPy/SciPy restatement of the SBROS-v1 path (TEST INFRASTRUCTURE).

Restates - it does not import - the reference's `SbrOS` algorithm
(/root/reference/gym_SBR/envs/gym_SBR_oneshot.py) for ONE environment, with the same
integrator the reference uses (SciPy's LSODA through `scipy.integrate.odeint`, an
un-vendored third-party dependency of the reference: call sites :1647, :1953, :2041,
:2318-2319, :2587) or, with `integrator="rk4"`, the fixed-step RK4 (10 substeps per control
interval) that the HIP kernels and the C oracle (oracle/sbr_oracle.c) use.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were captured from the
running reference by oracle/gen_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.

State vector: 0=V 1=Si 2=Ss 3=Xi 4=Xs 5=Xbh 6=Xba 7=Xp 8=So 9=Sno 10=Snh 11=Snd 12=Xnd 13=Salk
(gym_SBR_oneshot.py:185-188).
"""
import math

import numpy as np
from scipy.integrate import odeint

from . import sbr_params as P

# stoichiometric coefficients nu[component][process] (gym_SBR_oneshot.py:1689-1725), written with
# the same expressions so that the values are bit-identical
_NU = {
    (2, 1): -1 / P.YH, (5, 1): 1.0, (8, 1): -(1 - P.YH) / P.YH, (10, 1): -P.IXB, (13, 1): -P.IXB / 14,
    (2, 2): -1 / P.YH, (5, 2): 1.0, (9, 2): -((1 - P.YH) / (2.86 * P.YH)), (10, 2): -P.IXB,
    (13, 2): (1 - P.YH) / (14 * 2.86 * P.YH) - P.IXB / 14,
    (6, 3): 1.0, (8, 3): -(4.57 - P.YA) / P.YA, (9, 3): 1 / P.YA, (10, 3): -P.IXB - 1 / P.YA,
    (13, 3): -P.IXB / 14 - 1 / (7 * P.YA),
    (4, 4): 1 - P.IXP, (5, 4): -1.0, (7, 4): P.IXP, (12, 4): P.IXB - P.FP * P.IXP,
    (4, 5): 1 - P.IXP, (6, 5): -1.0, (7, 5): P.IXP, (12, 5): P.IXB - P.FP * P.IXP,
    (10, 6): 1.0, (11, 6): -1.0, (13, 6): 1 / 14,
    (2, 7): 1.0, (4, 7): -1.0,
    (11, 8): 1.0, (12, 8): -1.0,
}
# NOTE on naming: the reference's Spar = [Ya, Yh, fp, ixb, ixp] and its nu4_4 = 1 - Spar[4],
# nu7_4 = Spar[4], nu12_4 = Spar[3] - Spar[2]*Spar[4]  (gym_SBR_oneshot.py:1707-1715).


def process_rates(x):
    """The eight ASM1 process rates, gym_SBR_oneshot.py:1660-1685 (same association order)."""
    ss, xs, xbh, xba, so, sno, snh, snd, xnd = x[2], x[4], x[5], x[6], x[8], x[9], x[10], x[11], x[12]
    r1 = P.MUH * (ss / (P.KS + ss)) * (so / (P.KOH + so)) * xbh
    r2 = P.MUH * (ss / (P.KS + ss)) * (P.KOH / (so + P.KOH)) * (sno / (P.KNO + sno)) * P.ETAG * xbh
    r3 = P.MUA * (snh / (P.KNH + snh)) * (so / (P.KOA + so)) * xba
    r4 = P.BH * xbh
    r5 = P.BA * xba
    r6 = P.KA * snd * xbh
    r7 = P.KH * ((xs / xbh) / (P.KX + (xs / xbh))) * (
        (so / (P.KOH + so)) + P.ETAH * (P.KOH / (so + P.KOH)) * (sno / (P.KNO + sno))) * xbh
    r8 = (xnd / xs) * r7
    return r1, r2, r3, r4, r5, r6, r7, r8


def conversion(x, kla):
    """Net conversion rates r[1..13] incl. aeration, gym_SBR_oneshot.py:1731-1755."""
    rho = (None,) + process_rates(x)
    n = _NU
    r = [0.0] * 14
    r[2] = n[2, 1] * rho[1] + n[2, 2] * rho[2] + n[2, 7] * rho[7]
    r[4] = n[4, 4] * rho[4] + n[4, 5] * rho[5] + n[4, 7] * rho[7]
    r[5] = n[5, 1] * rho[1] + n[5, 2] * rho[2] + n[5, 4] * rho[4]
    r[6] = n[6, 3] * rho[3] + n[6, 5] * rho[5]
    r[7] = n[7, 4] * rho[4] + n[7, 5] * rho[5]
    r[8] = n[8, 1] * rho[1] + n[8, 3] * rho[3] + kla * (P.SO_SAT - x[8])
    r[9] = n[9, 2] * rho[2] + n[9, 3] * rho[3]
    r[10] = n[10, 1] * rho[1] + n[10, 2] * rho[2] + n[10, 3] * rho[3] + n[10, 6] * rho[6]
    r[11] = n[11, 6] * rho[6] + n[11, 8] * rho[8]
    r[12] = n[12, 4] * rho[4] + n[12, 5] * rho[5] + n[12, 8] * rho[8]
    r[13] = n[13, 1] * rho[1] + n[13, 2] * rho[2] + n[13, 3] * rho[3] + n[13, 6] * rho[6]
    return r


def rhs_reaction(x, t, kla, ec):
    """reaction_dxdt, gym_SBR_oneshot.py:1658-1787: conversion + carbon dosing/dilution."""
    r = conversion(x, kla)
    d = np.zeros(14)
    d[0] = 0 + ec
    q = ec / x[0]
    for i in range(1, 14):
        d[i] = r[i] + q * ((P.EC_CONC - x[i]) if i == 2 else (-x[i]))
    return d


def rhs_fill(x, t, kla, loading):
    """filling_dxdt with EC = 0, gym_SBR_oneshot.py:1424-1583.  The path only calls it with EC = 0,
    where the in-place dilution block :1523-1551 reduces to x[i] = (x[i]*V)/V: the identity
    mathematically but NOT bitwise (<= 1 ulp).  It is deliberately not restated; the known-answer
    test bounds the effect (3e-15 relative) and shows that adding it back gives bit equality."""
    r = conversion(x, kla)
    d = np.zeros(14)
    d[0] = loading[0]
    q = loading[0] / x[0]
    for i in range(1, 14):
        d[i] = r[i] + q * (loading[i] - x[i])
    return d


def rhs_fill_reference(x, t, kla, loading):
    """filling_dxdt exactly as the reference executes it at EC = 0 (gym_SBR_oneshot.py:1424-1583):
    rates from x, then the in-place block :1523-1551 overwrites x[i] with (x[i]*V)/V - which also
    mutates the integrator's working vector, as in the reference - then the loading terms.  With this
    form layer 1 in LSODA mode is BIT-IDENTICAL to the reference (tests/test_oracle_golden.py)."""
    r = conversion(x, kla)
    ec = 0.0
    x[0] = x[0] + ec
    for i in range(1, 14):
        x[i] = (x[i] * x[0] + P.EC_CONC * ec) / (x[0] + ec) if i == 2 else x[i] * x[0] / (x[0] + ec)
    d = np.zeros(14)
    d[0] = loading[0]
    for i in range(1, 14):
        d[i] = r[i] + (loading[0] / x[0]) * (loading[i] - x[i])
    return d


def rhs_idle(x, t, kla):
    """idle_dxdt, gym_SBR_oneshot.py:2424-2552: conversion only, volume constant."""
    r = conversion(x, kla)
    d = np.zeros(14)
    d[1:] = r[1:]
    return d


def rk4(f, x, t0, t1, n, args):
    """Classical RK4 with n equal substeps (what the HIP kernels do); f is autonomous here."""
    h = (t1 - t0) / n
    x = np.array(x, dtype=np.float64)
    for _ in range(n):
        k1 = f(x, 0.0, *args)
        k2 = f(x + (0.5 * h) * k1, 0.0, *args)
        k3 = f(x + (0.5 * h) * k2, 0.0, *args)
        k4 = f(x + h * k3, 0.0, *args)
        x = x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return x


def rhs_reaction_w(y, v0, kla, ec):
    """reaction_dxdt (gym_SBR_oneshot.py:1658-1787) in scaled-mass variables y = (V, w), w_i = c_i V/V0: substituting
    c = w/s, s = V/V0, into V' = ec, c_i' = r_i(c) + (ec/V)(c_in,i - c_i) removes the dilution terms identically:
    w_i' = s r_i(w/s) + (ec/V0) c_in,i with c_in = EC_conc for Ss, 0 otherwise.  Same operations, in the same order, as
    rhs_reaction_w of oracle/sbr_oracle.c."""
    s = y[0] / v0
    c = [y[0]] + [y[i] / s for i in range(1, 14)]
    r = conversion(c, kla)
    q0 = ec / v0
    d = np.zeros(14)
    d[0] = 0 + ec
    for i in range(1, 14):
        d[i] = s * r[i] + q0 * (P.EC_CONC if i == 2 else 0.0)
    return d


def rk4_reaction_w(x, t0, t1, n, kla, ec):
    """A carbon-dosing reaction interval (ec != 0) by classical RK4 on the scaled-mass system (round 4): what the HIP
    kernels' dosing loop integrates (sbr_device.h, sbr_rk4_dose) and, bit for bit, what oracle/sbr_oracle.c rk4_reaction_w
    does.  w = c at the start of the interval (s = 1), c = w/s at its end."""
    h = (t1 - t0) / n
    x = np.array(x, dtype=np.float64)
    v0 = x[0]
    for _ in range(n):
        k1 = rhs_reaction_w(x, v0, kla, ec)
        k2 = rhs_reaction_w(x + (0.5 * h) * k1, v0, kla, ec)
        k3 = rhs_reaction_w(x + (0.5 * h) * k2, v0, kla, ec)
        k4 = rhs_reaction_w(x + h * k3, v0, kla, ec)
        x = x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    s_end = x[0] / v0
    x[1:] = x[1:] / s_end
    return x


# ---------------------------------------------------------------------------------------------------------------------
# scheme 1 ("B5A", round 5): a control interval by Butcher's six-stage fifth-order Runge-Kutta scheme with a step count
# chosen PER INTERVAL from the plant's own state, instead of ten classical RK4 substeps.  Still an explicit fixed-formula
# integration of the reference's reaction_dxdt (gym_SBR_oneshot.py:1658-1787) with Kla and EC held; what changes is where
# the right-hand-side evaluations are spent (scripts/analysis/rhs_study.py):
#   * the only stiff mode of the system is the relaxation of dissolved oxygen,
#         lam(So) = a1 K_OH/(K_OH+So)^2 + a3 K_OA/(K_OA+So)^2 + Kla,   a1, a3 from rho1, rho3 of :1660-1668;
#     every other mode has |lambda| t_delta <= ~1.3, which ONE fifth-order step resolves to 1e-7;
#   * "slaved" intervals - |So| < 1e-9 and an aeration that could not lift it above that within the interval, i.e. the anoxic
#     phases once the oxygen is used up (So is 1e-10 ... 1e-52 in the reference there) - hold So during the steps and damp it
#     afterwards by 1/(1 + lam(0) span): TWO steps (one step's local error, <= 0.035 of the gate, is amplified past the
#     gate by the NO3-PID -> dosing loop in the golden episode random_b; two steps: 0.0008);
#   * otherwise n = 1, 2 or 4 steps from z = lam(So_lo) span, So_lo = the lowest So a linear projection over the interval
#     reaches (consumption slows as So falls, so the projection bounds So from below and z from above), with a1 and a3 taken at
#     the projected upper ends of Ss and Snh (carbon dosing raises Ss by up to 3 g/m3 within an interval);
#   * in the last case - the knee, where So moves through K_OH - n = max(4, floor(lam(0) span / 2.5) + 1): the worst-case
#     lam(0) h stays below 2.5 (Butcher-5 is stable on the real axis up to 3.39; the reference plant needs 4, 5 in `zeros`);
#   * and never fewer steps than the OTHER Monod arguments ask for: zs = max |slope_i| span / (K_i + |x_i|) over Ss, Snh, Sno
#     below 0.15 -> 1, below 0.5 -> 2, else 4 (no effect on any reference-captured interval; it matters for plants driven far
#     from the reference's regime, scripts/analysis/plan_probe.py);
#   * the idle phase of the done call (:2554-2597, one odeint over ~464 dt) is cut into ceil(rows/10) macro intervals, each
#     planned like a control interval.  The FILL phase (:1585-1654) stays with RK4 under either scheme: a fill interval cannot be
#     planned from its start state - the inflow raises Ss and Snh severalfold within the interval, and the oxygen uptake with
#     them (planned on its own, a fill interval of SBR-v2 was up to 70 gates off: scripts/analysis/cycle_intervals.py).
# oracle/sbr_oracle.c b5a_interval does the same operations in the same order (bit-identical, tests/test_oracle_golden.py).
B5A_SO_SLAVED, B5A_Z1, B5A_Z2, B5A_Z_STAB, B5A_N_MAX = 1e-9, 0.3, 1.0, 2.5, 64
B5A_ZS1, B5A_ZS2 = 0.15, 0.5
_A21, _A31, _A32, _A42, _A43, _A51, _A54 = 0.25, 0.125, 0.125, -0.5, 1.0, 3.0 / 16.0, 9.0 / 16.0
_A61, _A62, _A63, _A64, _A65 = -3.0 / 7.0, 2.0 / 7.0, 12.0 / 7.0, -12.0 / 7.0, 8.0 / 7.0
_B1, _B3, _B4, _B5, _B6 = 7.0 / 90.0, 32.0 / 90.0, 12.0 / 90.0, 32.0 / 90.0, 7.0 / 90.0


def b5a_plan(x, k1, span, kla):
    """(n, slaved, lam0): the step count of one macro interval from its start state x, the first stage slope k1 = f(x) (which
    does not depend on the step size), the span and Kla.  Everything the oxygen rate is built from is taken at its UPPER bound
    over the interval under a linear projection of the slow variables: Ss and Snh at max(start, start + slope span) (carbon
    dosing raises Ss within an interval), So at the lowest value the steeper of (its start slope, its slope with the projected
    substrate levels) reaches."""
    ss, xbh, xba, so, sno, snh = x[2], x[5], x[6], x[8], x[9], x[10]
    p2 = ss + k1[2] * span
    ss_hi = p2 if p2 > ss else ss
    p10 = snh + k1[10] * span
    snh_hi = p10 if p10 > snh else snh
    c1 = ((1 - P.YH) / P.YH) * P.MUH * xbh
    c3 = ((4.57 - P.YA) / P.YA) * P.MUA * xba
    m1s, m3s = ss / (P.KS + ss), snh / (P.KNH + snh)
    m1, m3 = ss_hi / (P.KS + ss_hi), snh_hi / (P.KNH + snh_hi)
    a1, a3 = c1 * m1, c3 * m3

    def lam(s):
        return a1 * P.KOH / ((P.KOH + s) * (P.KOH + s)) + a3 * P.KOA / ((P.KOA + s) * (P.KOA + s)) + kla
    slaved = (abs(so) < B5A_SO_SLAVED) and (kla * P.SO_SAT * span < B5A_SO_SLAVED)
    slope_hi = k1[8] - c1 * (m1 - m1s) * (so / (P.KOH + so)) - c3 * (m3 - m3s) * (so / (P.KOA + so))
    slope = slope_hi if slope_hi < k1[8] else k1[8]
    proj = so + slope * span
    lo1 = proj if proj < so else so
    so_lo = lo1 if lo1 > 0.0 else 0.0
    z_ub = lam(so_lo) * span
    lam0 = lam(0.0)
    if slaved:
        n = 2
    elif z_ub < B5A_Z1:
        n = 1
    elif z_ub < B5A_Z2:
        n = 2
    else:                                   # the knee: at least four steps, and lam(0) h <= Z_STAB (Butcher-5: stable to 3.39)
        q = lam0 * span / B5A_Z_STAB
        n = 4 if q < 4.0 else (B5A_N_MAX if not (q < float(B5A_N_MAX)) else int(q) + 1)
        # ... for a state inside the model's domain, where a1 and a3 are bounded (Monod factors in [0, 1]); outside (Ss or Snh negative
        # towards or beyond its pole, or NaN) the state is garbage and the count stays at the knee's four (see oracle/sbr_oracle.c)
        if not (abs(m1 - 0.5) <= 0.5 and abs(m3 - 0.5) <= 0.5):
            n = 4
    # how far the arguments of the other Monod terms move within the interval: |slope| span / (K + |x|)
    zs = abs(k1[2]) * span / (P.KS + abs(ss))
    z10 = abs(k1[10]) * span / (P.KNH + abs(snh))
    z9 = abs(k1[9]) * span / (P.KNO + abs(sno))
    zs = z10 if z10 > zs else zs
    zs = z9 if z9 > zs else zs
    n_s = 1 if zs < B5A_ZS1 else (2 if zs < B5A_ZS2 else 4)
    return (n_s if n_s > n else n), slaved, lam0


def b5_step(f, x, h, k1, hold_so):
    """One step of Butcher's fifth-order scheme (six stages; J. C. Butcher 1964); k1 = f(x) is passed in.  hold_so: the
    slope of component 8 is replaced by 0 in every stage."""
    def ev(y):
        k = f(y)
        if hold_so:
            k[8] = 0.0
        return k
    k2 = ev(x + (h * _A21) * k1)
    k3 = ev(x + (h * _A31) * k1 + (h * _A32) * k2)
    k4 = ev(x + (h * _A42) * k2 + (h * _A43) * k3)
    k5 = ev(x + (h * _A51) * k1 + (h * _A54) * k4)
    k6 = ev(x + (h * _A61) * k1 + (h * _A62) * k2 + (h * _A63) * k3 + (h * _A64) * k4 + (h * _A65) * k5)
    return x + (h * _B1) * k1 + (h * _B3) * k3 + (h * _B4) * k4 + (h * _B5) * k5 + (h * _B6) * k6


def b5a_macro(kind, x, span, kla, ec=0.0):
    """One macro interval of scheme 1.  kind 0: reaction (ec != 0: scaled-mass form), 2: idle.  Returns (x_end, n)."""
    x = np.array(x, dtype=np.float64)
    v0 = x[0]
    dose = kind == 0 and ec != 0
    if kind == 2:
        f = lambda y: rhs_idle(y, 0.0, kla)                 # noqa: E731
    elif dose:
        f = lambda y: rhs_reaction_w(y, v0, kla, ec)        # noqa: E731   scaled-mass variables, as rk4_reaction_w
    else:
        f = lambda y: rhs_reaction(y, 0.0, kla, ec)         # noqa: E731
    k1 = f(x)
    n, slaved, lam0 = b5a_plan(x, k1, span, kla)
    h = span / n
    if slaved:
        k1[8] = 0.0
    for s in range(n):
        if s > 0:
            k1 = f(x)
            if slaved:
                k1[8] = 0.0
        x = b5_step(f, x, h, k1, slaved)
    if dose:
        s_end = x[0] / v0
        x[1:] = x[1:] / s_end
    if slaved:
        x[8] = x[8] / (1.0 + lam0 * span)
    return x, n


def b5a_span(kind, x, span, m, kla, ec=0.0):
    """m macro intervals of span/m each (a control interval: m = 1; the idle phase: ceil(rows/10)).  Returns (x_end, n of
    the last macro interval)."""
    hm = span / m
    n = 0
    for _ in range(m):
        x, n = b5a_macro(kind, x, hm, kla, ec)
    return x, n


def influent_mix(means, stds, rnd):
    """buffer_tank3.py:68-107 for one scenario: series = mean + std*rnd (one rnd vector shared by
    all series), flow-weighted means; returns [0.66, Si..Salk].  means/stds: [14,48], last row = q."""
    series = means + stds * rnd[None, :]
    q = series[13]
    sq = sum(q)
    return np.array([0.66] + [sum(series[j] * q) / sq for j in range(13)])


def settle_closed_form(xf, c_t):
    """Layer concentrations after the settling phase.

    settling_dsXdt (gym_SBR_oneshot.py:2171-2262) has v = max(vmax, exp-exp) == vmax = 474
    always, hence the linear system dsX0 = c*sX1, dsXi = c*(sX(i+1) - sXi) (i=1..8),
    dsX9 = -c*sX9 with c = 474/z and all layers starting at Xf.  Exact solution with a = c*T:
    sX(9-j) = Xf*e^-a*sum_{m<=j} a^m/m!  (j = 0..8),  sX0 = 10*Xf - sum(others).
    """
    e = math.exp(-c_t)
    sx = np.zeros(10)
    term, partial = 1.0, 0.0
    for j in range(9):
        partial += term                  # sum_{m<=j} a^m/m!
        sx[9 - j] = xf * e * partial
        term *= c_t / (j + 1)
    sx[0] = 10.0 * xf - sum(sx[1:])       # sequential, the order oracle/sbr_oracle.c uses
    return sx


class SbrOsRef:
    """One SBROS-v1 environment, restated.  API mirrors SbrOS: reset() -> (obs_DO, obs_EC),
    step(a) -> ((obs_DO, obs_EC), state, reward, done, {})."""

    def __init__(self, tables=None, integrator="lsoda", settle="lsoda", scheme=1):
        self.tables = tables            # (means[8,14,48], stds[8,14,48]) or None if influent is given
        self.integrator = integrator
        self.settle = settle if integrator == "lsoda" else "closed"
        self.scheme = scheme            # "rk4" mode only: 0 = RK4 x n_sub per interval, 1 = adaptive Butcher-5 (b5a_reaction)
        self.step_counts = []           # scheme 1: the step count of every reaction interval (0 = fell back to RK4)

    # ------------------------------------------------------------------ integrate one span
    def _integrate(self, f, x, t0, t1, n_rows, n_sub, args):
        if self.integrator == "lsoda":
            grid = np.linspace(t0, t1, n_rows)
            rows = odeint(f, x, grid, args=args)
            return rows[-1].copy(), rows
        if f is rhs_reaction and self.scheme == 1:
            x1, n = b5a_span(0, x, t1 - t0, 1, *args)
            self.step_counts.append(n)
            return x1, None
        if f is rhs_idle and self.scheme == 1:
            return b5a_span(2, x, t1 - t0, (n_sub + 9) // 10, args[0])[0], None
        if f is rhs_reaction and args[1] != 0:          # a dosing interval: RK4 on the scaled-mass system (rk4_reaction_w)
            return rk4_reaction_w(x, t0, t1, n_sub, *args), None
        return rk4(f, x, t0, t1, n_sub, args), None

    # ------------------------------------------------------------------ reset (:168-438)
    def reset(self, rnd=None, scenario=P.SCENARIO_DEFAULT, influent=None):
        if influent is None:
            means, stds = self.tables
            influent = influent_mix(means[scenario], stds[scenario], np.asarray(rnd, dtype=np.float64))
        infl = np.array(influent, dtype=np.float64)
        self.iv = P.IV_INIT
        self.qin = P.WV - self.iv
        infl[0] = self.qin / P.T1_END                       # :287
        self.influent = infl
        x0 = np.array(P.X0_INIT, dtype=np.float64)
        self.u_do, self.u_ec = P.U_DO_INIT, P.U_EC_INIT
        # controller memories (lists in the reference; only the tails matter)
        so_hist = [x0[8]]
        sno_hist = [x0[9]]
        # Sim_filling :1585-1654 : DO-PID with sp=0 at t_start == 0  => ie = 0, dcv = 0
        e = 0 - so_hist[-1]
        ie_do = 0.0
        kla = P.KC_DO * e + P.KC_DO / P.TAUI_DO * ie_do + P.KC_DO * P.TAUD_DO * 0 + 0
        if kla > P.KLA_MAX:
            kla = P.KLA_MAX
            ie_do = ie_do - e * P.DT
        if kla < P.KLA_MIN:
            kla = P.KLA_MIN
            ie_do = ie_do - e * P.DT
        ie_ec = 0.0                                          # EC forced to 0, no clamp fires
        t_end = 0 + P.T_RATIO[0] * 0.5
        n_rows = int((t_end - 0) / P.DT)
        # LSODA mode restates the reference bit for bit; RK4 mode uses the clean form the kernels use
        f_fill = rhs_fill_reference if self.integrator == "lsoda" else rhs_fill
        x1, rows = self._integrate(f_fill, x0.copy(), 0.0, t_end, n_rows, n_rows, (kla, infl))
        so_hist.append(x1[8])
        sno_hist.append(x1[2])                               # :1652 stores Ss into the Sno memory
        self.x = x1
        self.t = t_end
        self.so_m1, self.so_m2 = so_hist[-1], so_hist[-2]
        self.sno_m1, self.sno_m2 = sno_hist[-1], sno_hist[-2]
        self.ie_do, self.ie_ec = ie_do, ie_ec
        # Kla/EC lists are [0, k] replicated (:320-324); keep the last 10 entries
        reps = n_rows // 2
        self.kla_hist = ([0.0, kla] * reps)[-10:]
        self.ec_prev = 0.0
        self.kla_last, self.ec_last = self.kla_hist[-1], 0.0
        self.done = False
        self.n_fill_rows = n_rows
        self.x_fill_rows = rows
        # reset observation: volume blend of influent and post-fill state (:346-361)
        blend = lambda i: (self.qin * infl[i] + x1[i] * self.iv) / (self.qin + self.iv)
        obs_do = [self.t / P.X1_DO[0]] + [blend(i) / s for i, s in zip(P.OBS_IDX_DO[1:], P.X1_DO[1:])]
        obs_ec = [self.t / P.X1_EC[0]] + [blend(i) / s for i, s in zip(P.OBS_IDX_EC[1:], P.X1_EC[1:])]
        return (obs_do + self._xdot(x0, x1, P.XDOT_DO), obs_ec + self._xdot(x0, x1, P.XDOT_EC))

    @staticmethod
    def _xdot(xa, xb, spec):
        return [min(1.0, max(-1.0, (xb[i] - xa[i]) / s)) for i, s in spec]

    # ------------------------------------------------------------------ one control interval
    def _interval(self, aerobic):
        """Sim_aero_rxn :1877-1963 / Sim_anaero_rxn :1965-2051 + run_*_step :1331-1419."""
        t0 = self.t
        t1 = t0 + P.T_DELTA
        n_rows = int((t1 - t0) / P.DT)                       # 9 or 10, fp-dependent (:1339, :1384)
        # DO PID (set-point 0 in anoxic intervals; output forced to 0 there but the integral winds)
        sp_do = self.u_do if aerobic else 0
        e = sp_do - self.so_m1
        dcv = (self.so_m1 - self.so_m2) / P.DT
        self.ie_do = self.ie_do + e * P.DT
        if aerobic:
            kla = P.KC_DO * e + P.KC_DO / P.TAUI_DO * self.ie_do + P.KC_DO * P.TAUD_DO * dcv + self.kla_last
        else:
            kla = 0
        if kla > P.KLA_MAX:
            kla = P.KLA_MAX
            self.ie_do = self.ie_do - e * P.DT
        if kla < P.KLA_MIN:
            kla = P.KLA_MIN
            self.ie_do = self.ie_do - e * P.DT
        # NO3 PID -> carbon dosing (error sign reversed; forced to 0 in aerobic intervals)
        e2 = self.sno_m1 - self.u_ec
        dcv2 = (self.sno_m1 - self.sno_m2) / P.DT
        self.ie_ec = self.ie_ec + e2 * P.DT
        if aerobic:
            ec = 0
        else:
            ec = P.KC_EC * e2 + P.KC_EC / P.TAUI_EC * self.ie_ec + P.KC_EC * P.TAUD_EC * dcv2 + self.ec_last
        if ec < P.EC_MIN:
            ec = P.EC_MIN
            self.ie_ec = self.ie_ec - e2 * P.DT
        elif ec > P.EC_MAX:
            ec = P.EC_MAX
            self.ie_ec = self.ie_ec - e2 * P.DT
        x0 = self.x
        x1, rows = self._integrate(rhs_reaction, x0, t0, t1, n_rows, 10, (float(kla), float(ec)))
        # bookkeeping of the lists the reward looks at
        self.kla_hist = (self.kla_hist + [float(kla)])[-10:]
        self.ec_prev, self.ec_last = self.ec_last, float(ec)
        self.kla_last = float(kla)
        self.so_m2, self.so_m1 = self.so_m1, x1[8]
        self.sno_m2, self.sno_m1 = self.sno_m1, x1[9]
        self.x_start, self.x, self.t = x0, x1, t1
        self.last = dict(t0=t0, t1=t1, n_rows=n_rows, kla=float(kla), ec=float(ec), rows=rows,
                         aerobic=aerobic, x_start=x0, x_end=x1)
        self.intervals.append(self.last)

    # ------------------------------------------------------------------ reward (module_reward_EQIOCI.py:4-115)
    def _reward(self):
        x = self.x
        n = self.last["n_rows"]
        xi, xs, xbh, xba, xp = x[3], x[4], x[5], x[6], x[7]
        snkj = x[10] + x[11] + x[12] + 0.08 * (xbh + xba) + 0.06 * (xp + xi)
        ss_ = 0.75 * (xs + xi + xbh + xba + xp)
        bod5 = 0.25 * (x[2] + xs + (1 - 0.08) * (xbh + xba))
        cod = x[2] + x[1] + xs + xi + xbh + xba + xp
        eqi = (2 * ss_ + 1 * cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1 / 1000) * 0.66
        eqi2 = eqi / 10
        td = 0.002 / 24
        span = self.last["t1"] - self.last["t0"]
        # Kla got ONE append per interval: Kla[-n:-1] = the n-1 values before the current one
        ae_dt = 1.32 * sum(self.kla_hist[-n:-1]) * td
        ae = 8 / (span * 1.8 * 1000) * ae_dt
        ae_max = 1.32 * (240 * 11) * td * (8 / ((td * 11) * 1.8 * 1000))
        # EC got n-1 appends per interval: EC[-n:-1] = last value of the previous interval + (n-2) current
        ec_sum = sum([self.ec_prev] + [self.ec_last] * (n - 2))
        ec_oci = P.EC_CONC * ec_sum * td / (span * 1000)
        ec_max = P.EC_CONC * (0.0005 * 11) * td / ((td * 11) * 1000)
        oci = ae + ec_oci
        reward = (1 - (eqi2 ** 2 + oci ** 2)) / 473
        self.reward_parts = (eqi2, ae / ae_max + ec_oci / ec_max, ae / ae_max, ec_oci / ec_max)
        return reward

    def _obs(self, t_obs, x, x_from):
        state = np.array([t_obs] + list(x)) / np.array(P.X1_STATE, dtype=np.float64)
        obs_do = [t_obs / P.X1_DO[0]] + [x[i] / s for i, s in zip(P.OBS_IDX_DO[1:], P.X1_DO[1:])]
        obs_ec = [t_obs / P.X1_EC[0]] + [x[i] / s for i, s in zip(P.OBS_IDX_EC[1:], P.X1_EC[1:])]
        return (obs_do + self._xdot(x_from, x, P.XDOT_DO), obs_ec + self._xdot(x_from, x, P.XDOT_EC)), state

    # ------------------------------------------------------------------ step (:843-1273)
    def step(self, action):
        self.intervals = []
        a_do = min(max(float(action[0]), 0.0), P.ACT_DO_MAX)
        a_ec = min(max(float(action[1]), 0.0), P.ACT_EC_MAX)
        # four sequential tests on the running time: a call that crosses a phase boundary runs
        # a second interval (3 times per episode)
        if self.t < P.T3_0:
            self.u_ec, self.u_do = a_ec, 0
            self._interval(aerobic=False)
        if (self.t >= P.T3_0) and (self.t <= P.T3_END):
            self.u_do, self.u_ec = a_do, 0
            self._interval(aerobic=True)
        if (self.t > P.T3_END) and (self.t <= P.T4_END):
            self.u_ec, self.u_do = a_ec, 0
            self._interval(aerobic=False)
        if self.t > P.T4_END:
            self.u_do, self.u_ec = a_do, 0
            self._interval(aerobic=True)
        reward = self._reward()
        obs, state = self._obs(self.t, self.x, self.x_start)
        done = False
        if self.t >= P.T5_END:
            done = True
            x_pre = self.x
            x_drawn = self._settle_draw(x_pre, self.t)
            x_idle = self._idle(x_drawn)
            obs, state = self._obs(P.T_CYCLE, x_idle, x_pre)
            self.x_after_draw, self.x_after_idle = x_drawn, x_idle
            self.done = True
        return obs, state, reward, done, {}

    # ------------------------------------------------------------------ settle + draw (:2264-2420)
    def _settle_draw(self, x, t):
        xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7])
        vs = x[0]
        z = vs / P.SETTLER_AREA
        t_set = P.T_RATIO[5] * P.T_CYCLE
        if self.settle == "lsoda":
            def f(sx, tt):
                j = P.SETTLER_VMAX * sx                       # v == vmax always (:2226-2235)
                d = np.empty(10)
                d[0] = j[1] / z
                d[1:9] = (j[2:10] - j[1:9]) / z
                d[9] = (0 - j[9]) / z
                return d
            grid = np.linspace(t, t + t_set, int(t_set / P.T_DELTA))
            sx = odeint(f, [xf] * 10, grid)[-1]
        else:
            sx = settle_closed_form(xf, P.SETTLER_VMAX / z * t_set)
        self.t_after_draw = (t + t_set) + P.T_RATIO[6] * P.T_CYCLE
        layer_v = vs / 10
        resid_v = vs - P.QEFF
        m = int(math.ceil(round(P.QEFF / layer_v)))
        self.sx_eff = sum(sx[-m:-1] * layer_v)                # drops the last layer (:2344)
        w = layer_v * sx[0:10 - m]
        resid_sx = sx[0:10 - m].copy()
        waste = sum(w) - P.BIOMASS_SETPOINT * resid_v
        qw = float("nan")
        for i in range(10 - m):
            rest = waste - w[i]
            if rest > 0:
                waste = rest
                resid_sx[i] = 0
                w[i] = 0
                resid_v -= layer_v
            else:
                qw = waste / (resid_sx[i] - P.BIOMASS_SETPOINT)
                w[i] = w[i] - qw * resid_sx[i]
                resid_v -= qw
                resid_sx[i] = w[i] / (layer_v - qw)
                break
        self.qw, self.waste_w = qw, waste
        sx2 = sum(w) / resid_v
        xn = np.array(x, dtype=np.float64)
        xn[0] = resid_v
        for i in (3, 4, 5, 6, 7):
            xn[i] = x[i] * (1 / 0.75) * sx2 / xf
        return xn

    # ------------------------------------------------------------------ idle (:2554-2597)
    def _idle(self, x):
        t0 = self.t_after_draw
        t1 = P.T_CYCLE
        # So memory was extended with the (constant) settle/draw rows: So[-1] == So[-2] == x[8]
        e = self.u_do - x[8]
        self.ie_do = self.ie_do + e * P.DT
        kla = P.KC_DO * e + P.KC_DO / P.TAUI_DO * self.ie_do + P.KC_DO * P.TAUD_DO * 0.0 + self.kla_last
        if kla > P.KLA_MAX:
            kla = P.KLA_MAX
            self.ie_do = self.ie_do - e * P.DT
        if kla < P.KLA_MIN:
            kla = P.KLA_MIN
            self.ie_do = self.ie_do - e * P.DT
        self.kla_idle = float(kla)
        n_rows = int((t1 - t0) / P.DT)
        x1, _ = self._integrate(rhs_idle, x, t0, t1, n_rows, n_rows, (float(kla),))
        self.n_idle_rows = n_rows
        return x1
