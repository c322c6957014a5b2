"""Round 6, VERDICT r5 item 2: one bounded, oracle-side attempt at the oxygen knee - can the aerobic wave-calls of the bench's
workload come down from four Butcher-5 steps to three or fewer without losing the parity gate?  Test infrastructure (numpy + the C
oracle + the committed fixtures), no GPU.

    python scripts/analysis/knee_study.py populations     builds /tmp/knee_pop.npz: the reference-captured intervals, the bench
                                                          workload's aerobic intervals (C oracle, physical policy, 16 384 envs)
                                                          and the idle phase's macro intervals of its done calls
    python scripts/analysis/knee_study.py knee            error of Butcher-5 x n on the knee intervals, by direction and z
    python scripts/analysis/knee_study.py waves           what candidate rules would do to the steps per WAVEFRONT
    python scripts/analysis/knee_study.py rosenbrock      a linearly implicit treatment of the scalar oxygen equation in the stages
    python scripts/analysis/knee_study.py idle            the idle phase of the done call: its plan, and a quasi-steady refinement

The numbers this prints are in profiles/r06_notes.md section 2."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "analysis"))
import rhs_study as S  # noqa: E402
from oracle import sbr_params as P  # noqa: E402

POP = os.environ.get("KNEE_POP", "/tmp/knee_pop.npz")
Z1, Z2, Z_STAB, ZS1, ZS2, SO_SLAVED = 0.3, 1.0, 2.5, 0.15, 0.5, 1e-9


def plan(x, span, kla, ec):
    """The shipped plan (oracle/sbr_ref.py b5a_plan), vectorised over intervals: x [14, N].  Returns a dict of its pieces."""
    v0 = x[0]
    k1 = np.where(ec != 0, S.f_w(x, v0, kla, ec), S.f(x, kla, ec))
    ss, xbh, xba, so, sno, snh = x[2], x[5], x[6], x[8], x[9], x[10]
    ss_hi = np.maximum(ss, ss + k1[2] * span)
    snh_hi = np.maximum(snh, snh + k1[10] * span)
    c1 = ((1 - P.YH) / P.YH) * P.MUH * xbh
    c3 = ((4.57 - P.YA) / P.YA) * P.MUA * xba
    m1s, m3s = ss / (P.KS + ss), snh / (P.KNH + snh)
    m1, m3 = ss_hi / (P.KS + ss_hi), snh_hi / (P.KNH + snh_hi)
    a1, a3 = c1 * m1, c3 * m3
    lam = lambda s: a1 * P.KOH / ((P.KOH + s) ** 2) + a3 * P.KOA / ((P.KOA + s) ** 2) + kla      # noqa: E731
    slaved = (np.abs(so) < SO_SLAVED) & (kla * P.SO_SAT * span < SO_SLAVED)
    slope_hi = k1[8] - c1 * (m1 - m1s) * (so / (P.KOH + so)) - c3 * (m3 - m3s) * (so / (P.KOA + so))
    slope = np.minimum(slope_hi, k1[8])
    so_lo = np.maximum(0.0, np.minimum(so, so + slope * span))
    z_ub, lam0 = lam(so_lo) * span, lam(0.0)
    q = lam0 * span / Z_STAB
    n_knee = np.where(q < 4.0, 4, np.minimum(64, q.astype(np.int64) + 1))
    n = np.where(slaved, 2, np.where(z_ub < Z1, 1, np.where(z_ub < Z2, 2, n_knee)))
    zs = np.maximum.reduce([np.abs(k1[2]) * span / (P.KS + np.abs(ss)), np.abs(k1[10]) * span / (P.KNH + np.abs(snh)),
                            np.abs(k1[9]) * span / (P.KNO + np.abs(sno))])
    n_s = np.where(zs < ZS1, 1, np.where(zs < ZS2, 2, 4))
    return {"n": np.maximum(n, n_s), "n_z": n, "n_s": n_s, "slaved": slaved, "z": z_ub, "lam0": lam0, "lam_start": lam(so), "k1": k1,
            "so_lo": so_lo, "slope": slope, "slope_hi": slope_hi, "zs": zs}


def b5(x, span, n, kla, ec, hold=None, rhs_mod=None):
    """Butcher-5 x n (n an int), scaled-mass form where ec != 0, So held where `hold`; vectorised.  rhs_mod(k, y, h) may alter a slope."""
    dose = ec != 0
    v0 = x[0].copy()
    mask = np.ones_like(x)
    if hold is not None:
        mask[8, hold] = 0.0

    def rhs(y, kla_, ec_):
        k = np.where(dose, S.f_w(y, v0, kla_, ec_), S.f(y, kla_, ec_)) * mask
        return k
    y = S.integrate(S.B5, x, span, n, kla, ec, rhs=rhs)
    s_end = np.where(dose, y[0] / v0, 1.0)
    y[1:] = y[1:] / s_end
    return y


def exact(x, span, kla, ec, n=160):
    return S.integrate(S.RK4, x, span, n, kla, ec)


# ------------------------------------------------------------------------------------------------ populations
def populations():
    from gym_sbr2_amd.vec_env import load_influent_tables
    from oracle import sbr_oracle as O
    iv = S.load_intervals()
    out = {"ref_X": iv["X"], "ref_span": iv["span"], "ref_kla": iv["kla"], "ref_ec": iv["ec"], "ref_kind": iv["kind"]}
    means, stds = load_influent_tables()
    n = 16384
    b = O.OracleBatch(n, O.default_params(scheme=1), nthreads=len(os.sched_getaffinity(0)))
    scen = (4 + np.arange(n) % 4).astype(np.int32)
    b.reset(b.mix(means, stds, scen, b.normals(0)))
    rs = np.random.RandomState(0)
    X, span, kla, ec, call, plans = [], [], [], [], [], []
    per_call_plan = []
    for c in range(463):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)]).astype(np.float32).astype(np.float64)
        if c == 462:
            pre = b.envs.copy()
            p2 = O.default_params(scheme=1); p2.terminal = 0
            b2 = O.OracleBatch(n, p2, nthreads=b.nthreads); b2.envs[:] = pre
            b2.step(a, want_obs=False)
            out["done_x_pre_settle"] = b2.envs["x"].T.copy()
            out["done_u_do"], out["done_ie_do"], out["done_kla_last"], out["done_t"] = (b2.envs["u_do"].copy(), b2.envs["ie_do"].copy(),
                                                                                         b2.envs["kla_last"].copy(), b2.envs["t"].copy())
        b.step(a, want_obs=False)
        pl = b.envs["scheme_plan"] & 0xff
        per_call_plan.append(pl.astype(np.uint8).copy())
        aer = b.envs["kla_last"] > 0
        keep = np.arange(n)[::8]                       # the last interval of the call for every eighth env
        X.append(b.envs["x_start"][keep].T.copy()); span.append(b.envs["span"][keep].copy()); kla.append(b.envs["kla_last"][keep].copy())
        ec.append(b.envs["ec_last"][keep].copy()); call.append(np.full(len(keep), c)); plans.append(pl[keep].copy())
    out.update(bench_X=np.concatenate(X, axis=1), bench_span=np.concatenate(span), bench_kla=np.concatenate(kla), bench_ec=np.concatenate(ec),
               bench_call=np.concatenate(call), bench_plan=np.concatenate(plans), bench_plan_all=np.array(per_call_plan),
               done_x_after=b.envs["x"].T.copy(), done_qw=b.envs["qw"].copy())
    np.savez_compressed(POP, **out)
    print("wrote", POP, {k: v.shape for k, v in out.items()})


def _load():
    if not os.path.exists(POP):
        populations()
    return dict(np.load(POP))


def _knee_table(tag, x, span, kla, ec):
    pl = plan(x, span, kla, ec)
    knee = (pl["n_z"] >= 4) & ~pl["slaved"]
    x, span, kla, ec = x[:, knee], span[knee], kla[knee], ec[knee]
    z, lam0s = pl["z"][knee], (pl["lam0"] * 1.0)[knee] * span
    ex = exact(x, span, kla, ec)
    falling = pl["k1"][8][knee] < 0
    print("\n%s: %d knee intervals (plan n >= 4) of %d; So falling at the start in %d, rising in %d" % (tag, knee.sum(), len(knee), falling.sum(), (~falling).sum()))
    res = {}
    for n in (2, 3, 4, 5):
        with np.errstate(all="ignore"):
            g = S.gate(b5(x, span, n, kla, ec), ex)
        res[n] = np.where(np.isfinite(g), g, 1e30)
    edges = [1.0, 1.5, 3.0, 5.0, 8.0, 1e9]
    print("   z = lam(So_lo) span | direction | count | worst gate with n = 2 | n = 3 | n = 4 | n = 5 | share of n = 3 within 0.6")
    for lo, hi in zip(edges[:-1], edges[1:]):
        for name, d in (("falling", falling), ("rising", ~falling)):
            m = (z >= lo) & (z < hi) & d
            if m.sum():
                print("   %4.1f - %-6s | %-7s | %5d | %8.3g | %8.3g | %8.3g | %8.3g | %.4f" % (
                    lo, "inf" if hi > 1e8 else "%.1f" % hi, name, m.sum(), res[2][m].max(), res[3][m].max(), res[4][m].max(), res[5][m].max(),
                    (res[3][m] <= 0.6).mean()))
    print("   all knee intervals: n = 3 worst %.3g (within 0.6: %.4f), n = 4 worst %.3g; lam(0) span: median %.2f max %.2f (three steps stable "
          "only below 3 x 2.5 = 7.5: %.3f of them)" % (res[3].max(), (res[3] <= 0.6).mean(), res[4].max(), np.median(lam0s), lam0s.max(), (lam0s < 7.5).mean()))
    return pl, knee, res


def knee():
    d = _load()
    _knee_table("reference-captured intervals", d["ref_X"], d["ref_span"], d["ref_kla"], d["ref_ec"])
    _knee_table("bench workload (physical policy), every eighth env", d["bench_X"], d["bench_span"], d["bench_kla"], d["bench_ec"])


def waves():
    """Steps per wavefront (the slowest of 64 lanes) on the bench workload under candidate rules, from the oracle's per-call plans."""
    d = _load()
    pa = d["bench_plan_all"].astype(np.int64) & 127            # [463, 16384]
    sched_aer = (d["bench_plan_all"] & 128) == 0
    w = pa.reshape(463, -1, 64)
    lane, wave = pa.mean(), w.max(axis=2).mean()
    aer_calls = [c for c in range(463) if (pa[c] >= 4).mean() > 0.01]
    print("shipped plan: %.3f steps per env and interval, %.3f per wavefront; %d calls with > 1 %% of the lanes at >= 4 steps" % (lane, wave, len(aer_calls)))
    print("   share of lanes at >= 4 steps in those calls: %.4f; share of wavefronts with such a lane: %.4f" % (
        (pa[aer_calls] >= 4).mean(), (w[aer_calls].max(axis=2) >= 4).mean()))
    for name, cap in (("every knee lane at three steps (if accuracy and stability allowed it)", 3), ("... at two", 2)):
        q = np.minimum(pa, cap)
        q = np.where(pa > 4, pa, q)                           # lanes the stability rule pushes above four stay there
        wq = q.reshape(463, -1, 64).max(axis=2).mean()
        print("   %s: %.3f per wavefront (%.3f fewer): x ~1.0 us per step = %.2f us per average call (the go bar: 0.6)" % (name, wq, wave - wq, (wave - wq) * 1.0))
    # a rule that spares a FRACTION f of the knee lanes (whichever way) leaves the wavefront's count unchanged unless all its knee lanes are spared
    for f in (0.5, 0.9, 0.99):
        rs = np.random.RandomState(1)
        spared = rs.uniform(size=pa.shape) < f
        q = np.where((pa == 4) & spared, 3, pa)
        wq = q.reshape(463, -1, 64).max(axis=2).mean()
        print("   three steps for a random %.0f %% of the four-step lanes: %.3f per wavefront (%.3f fewer)" % (100 * f, wq, wave - wq))


def rosenbrock():
    """Candidate (b): the scalar oxygen equation treated linearly implicitly inside the explicit stages - every stage slope of So
    divided by (1 + gamma h lam(So_stage)) (a W-method's damping with the exact scalar Jacobian: one division per stage, no Newton
    loop).  Two and three steps on the knee intervals, against RK4 x 160."""
    d = _load()
    for tag in ("ref", "bench"):
        x, span, kla, ec = d[tag + "_X"], d[tag + "_span"], d[tag + "_kla"], d[tag + "_ec"]
        pl = plan(x, span, kla, ec)
        knee = (pl["n_z"] >= 4) & ~pl["slaved"] & (ec == 0)
        x, span, kla, ec = x[:, knee], span[knee], kla[knee], ec[knee]
        ex = exact(x, span, kla, ec)
        print("\n%s: %d knee intervals without dosing" % (tag, knee.sum()))
        for n in (2, 3):
            for gamma in (0.0, 0.25, 0.5):
                h = span / n

                def rhs(y, kla_, ec_, h=h, gamma=gamma):
                    k = S.f(y, kla_, ec_)
                    if gamma:
                        k[8] = k[8] / (1.0 + gamma * h * S.so_rate(y, kla_))
                    return k
                with np.errstate(all="ignore"):
                    g = S.gate(S.integrate(S.B5, x, span, n, kla, ec, rhs=rhs), ex)
                g = np.where(np.isfinite(g), g, 1e30)
                print("   n = %d, gamma = %.2f: worst %9.3g, p99 %9.3g, within 0.6: %.4f" % (n, gamma, g.max(), np.percentile(g, 99), (g <= 0.6).mean()))


def idle():
    """The idle phase of the done call on the bench workload: 47 macro intervals from the drawn reactor with Kla held.  What the
    shipped plan does there, and what a quasi-steady refinement would do: when dissolved oxygen sits at its quasi-steady state
    INSIDE the knee (aeration balancing uptake), z = lam(So) span says `knee' although nothing is moving."""
    from oracle import sbr_oracle as O
    d = _load()
    p = O.default_params(scheme=1)
    x = d["done_x_pre_settle"].copy()
    n_env = x.shape[1]
    # settle + draw + the idle phase's PID update, by the C oracle's own terminal() on a copy: take its outputs via one full done call
    # (d["done_x_after"]) and replay the idle phase here from the drawn reactor, macro interval by macro interval
    xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7])
    vs = x[0]
    a = p.settler_vmax / (vs / p.settler_area) * (p.t_settle * p.t_cycle)
    ea = np.exp(-a)
    sx = np.zeros((10, n_env)); term = np.ones(n_env); partial = np.zeros(n_env)
    for j in range(9):
        partial = partial + term
        sx[9 - j] = xf * ea * partial
        term = term * a / (j + 1)
    sx[0] = 10.0 * xf - sx[1:].sum(axis=0)
    layer_v = vs / 10
    m = 5
    w = layer_v * sx[:10 - m]
    waste = w.sum(axis=0) - p.biomass_setpoint * (vs - p.Qeff)
    qw = waste / (sx[0] - p.biomass_setpoint)                      # the reference regime: the first layer is the partially wasted one
    ok = (waste - w[0] <= 0) & (qw > 0)
    resid = vs - p.Qeff - qw
    wk = w.sum(axis=0) - qw * sx[0]
    xd = x.copy()
    xd[0] = resid
    xd[3:8] = x[3:8] * (1 / 0.75) * (wk / resid) / xf
    print("idle phase: %d envs, %d in the ordinary draw branch; Qw replay vs oracle max |d| %.2e" % (n_env, ok.sum(), np.abs(qw - d["done_qw"])[ok].max()))
    e = d["done_u_do"] - xd[8]
    ie = d["done_ie_do"] + e * p.dt
    kla = np.clip(p.Kc_DO * e + p.Kc_DO / p.tauI_DO * ie + d["done_kla_last"], p.Kla_min, p.Kla_max)
    t_after = (d["done_t"] + p.t_settle * p.t_cycle) + p.t_draw * p.t_cycle
    span = p.t_cycle - t_after
    rows = (span / p.dt).astype(int)
    mm = (rows[0] + 9) // 10
    hm = span / mm
    xs = xd[:, ok]; kl = kla[ok]; hh = hm[ok]; z0 = np.zeros(ok.sum())
    counts, stat = [], []
    x_ship = xs.copy()
    for j in range(mm):
        pl = plan(x_ship, hh, kl, z0)
        rel = np.abs(pl["k1"][8]) * hh / (P.KOH + np.abs(x_ship[8]))
        counts.append(pl["n"].copy()); stat.append(rel.copy())
        out = np.empty_like(x_ship)
        for nn in np.unique(pl["n"]):
            sel = pl["n"] == nn
            out[:, sel] = b5(x_ship[:, sel], hh[sel], int(nn), kl[sel], z0[sel], hold=pl["slaved"][sel])
        sl = pl["slaved"]
        out[8, sl] = out[8, sl] / (1.0 + pl["lam0"][sl] * hh[sl])
        x_ship = out
    counts, stat = np.array(counts), np.array(stat)
    print("   shipped plan: %d macro intervals, mean steps per env and macro interval %.2f; per WAVEFRONT (max of 64) %.2f -> %.0f steps per wavefront "
          "and done call" % (mm, counts.mean(), counts.reshape(mm, -1, 64).max(axis=2).mean(), counts.reshape(mm, -1, 64).max(axis=2).sum(axis=0).mean()))
    print("   share of (env, macro interval) at >= 4 steps: %.3f; of those, So's Monod argument moves by less than 5 %% within the interval "
          "(|So'| h / (K_OH + So) < 0.05): %.3f" % ((counts >= 4).mean(), (stat[counts >= 4] < 0.05).mean()))
    ex = exact(xs, span[ok], kl, z0, n=4000)
    print("   shipped plan vs RK4 x 4000 over the whole idle phase: worst gate %.3g" % S.gate(x_ship, ex).max())
    # the refinement: in the knee branch, when So is quasi-steady (rel < r0) take n = max(1 or 2, ceil(z / 2.5))
    for r0, nmin, zq, rt in ((0.01, 2, 2.5, None), (0.02, 2, 2.5, None), (0.05, 2, 2.5, None), (0.1, 2, 2.5, None), (0.2, 2, 2.5, None), (0.05, 2, 2.0, None)):
        print("   [tight tier below rel %s]" % rt, end="")
        xq = xs.copy(); cq = []
        for j in range(mm):
            pl = plan_qs(xq, hh, kl, z0, r0, nmin, zq, rt)
            nq = pl["n"]
            cq.append(nq.copy())
            out = np.empty_like(xq)
            for nn in np.unique(nq):
                sel = nq == nn
                out[:, sel] = b5(xq[:, sel], hh[sel], int(nn), kl[sel], z0[sel], hold=pl["slaved"][sel])
            sl = pl["slaved"]
            out[8, sl] = out[8, sl] / (1.0 + pl["lam0"][sl] * hh[sl])
            xq = out
        cq = np.array(cq)
        print("   quasi-steady rule r0 = %.2f, at least %d step(s), lam h <= %.2f: %.2f steps per env, %.1f per wavefront and done call (shipped %.1f); whole idle "
              "phase vs RK4 x 4000: worst gate %.3g" % (r0, nmin, zq, cq.mean(), cq.reshape(mm, -1, 64).max(axis=2).sum(axis=0).mean(),
                                                    counts.reshape(mm, -1, 64).max(axis=2).sum(axis=0).mean(), S.gate(xq, ex).max()))




def plan_qs(x, span, kla, ec, r0=0.05, nmin=2, zq=Z_STAB, r_tight=None, zq_loose=1.25):
    """The shipped plan + the quasi-steady refinement of round 6: in the knee branch, when dissolved oxygen is quasi-steady - its
    Monod argument moves by less than r0 over the interval at the steeper of its two start slopes - the count is what STABILITY asks
    for, max(nmin, ceil(z / 2.5)), instead of the transient's four."""
    pl = plan(x, span, kla, ec)
    so = x[8]
    k1so = pl["k1"][8]
    rel = np.maximum(np.abs(k1so), np.abs(pl["slope_hi"])) * span / (P.KOH + np.abs(so))
    stab = np.maximum(nmin, np.ceil(pl["z"] / zq)).astype(np.int64)
    if r_tight is not None:           # two tiers: lam h <= zq where So is all but stationary (rel < r_tight), <= zq_loose up to r0
        stab = np.where(rel < r_tight, stab, np.maximum(nmin, np.ceil(pl["z"] / zq_loose)).astype(np.int64))
    qs = (pl["n_z"] >= 4) & ~pl["slaved"] & (rel < r0)
    n_z = np.where(qs, np.minimum(stab, pl["n_z"]), pl["n_z"])
    pl2 = dict(pl, n=np.maximum(n_z, pl["n_s"]), qs=qs, rel=rel)
    return pl2


def _apply(x, span, kla, ec, pl):
    out = np.empty_like(x)
    for nn in np.unique(pl["n"]):
        sel = pl["n"] == nn
        out[:, sel] = b5(x[:, sel], span[sel], int(nn), kla[sel], ec[sel], hold=pl["slaved"][sel])
    sl = pl["slaved"]
    out[8, sl] = out[8, sl] / (1.0 + pl["lam0"][sl] * span[sl])
    return out


def qs():
    """Open loop: the quasi-steady refinement on the reference-captured intervals (against the reference's own LSODA end states and
    against RK4 x 160) and on the bench workload's intervals."""
    d = _load()
    for tag in ("ref", "bench"):
        x, span, kla, ec = d[tag + "_X"], d[tag + "_span"], d[tag + "_kla"], d[tag + "_ec"]
        if tag == "bench":
            x, span, kla, ec = x[:, ::4], span[::4], kla[::4], ec[::4]
        ex = exact(x, span, kla, ec)
        base = plan(x, span, kla, ec)
        g0 = S.gate(_apply(x, span, kla, ec, base), ex)
        print("\n%s: %d intervals; shipped plan: %.3f steps per interval, worst gate vs RK4 x 160 %.3g" % (tag, len(span), base["n"].mean(), g0.max()))
        for r0, nmin, zq, rt in ((0.02, 2, 2.5, None), (0.1, 2, 2.5, 0.01), (0.1, 2, 2.5, 0.005), (0.05, 2, 2.5, 0.01), (0.1, 2, 2.0, 0.01)):
            if True:
                pl = plan_qs(x, span, kla, ec, r0, nmin, zq, rt)
                print("   [tight tier below rel %s]" % rt, end="")
                g = S.gate(_apply(x, span, kla, ec, pl), ex)
                ch = pl["n"] != base["n"]
                print("   r0 = %.2f, at least %d, lam h <= %.2f: %.3f steps per interval; %5d intervals change count (%.4f of the >= 4-step ones); worst gate %.3g, "
                      "worst over the changed ones %.3g" % (r0, nmin, zq, pl["n"].mean(), ch.sum(), ch.sum() / max((base["n"] >= 4).sum(), 1), g.max(),
                                                            g[ch].max() if ch.any() else 0.0))


if __name__ == "__main__":
    {"populations": populations, "knee": knee, "waves": waves, "rosenbrock": rosenbrock, "idle": idle, "qs": qs}[sys.argv[1] if len(sys.argv) > 1 else "knee"]()
