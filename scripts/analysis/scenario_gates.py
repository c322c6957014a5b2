"""Per-scenario parity figures of DESIGN.md section 4.3 (round 5), from the committed fixtures and the CPU oracle only.

For every reference-captured episode (tests/golden/sbros_*.npz: the six scenario-6 episodes and the 18 scenario episodes of
oracle/gen_golden.py scenario_cases) on the calls where parity is defined (tests/conftest.py valid_calls):

  open      worst gate of ONE interval (scheme 0: RK4 x 10; `_s1`: cfg.scheme = 1) started from the reference's own state,
            against the reference's end state
  closed    worst gate of the chained episode against the reference run at odeint rtol = atol = 1e-12 (both schemes)
  default   the same against the reference at its default tolerance (carries LSODA's own noise, section 4.3)
  lam_dt    largest |eigenvalue| of the right-hand side's Jacobian (central differences, the interval's Kla and EC) at the
            interval start states, times h = dt; classical RK4 is stable on the negative real axis up to 2.785

    python scripts/analysis/scenario_gates.py            prints the table and writes profiles/r05_scenario_gates.json
    python scripts/analysis/scenario_gates.py --write    also replaces the table between the markers of DESIGN.md
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EPISODES, SCENARIO_EPISODES, gate, golden, valid_calls  # noqa: E402
from oracle import sbr_oracle as O  # noqa: E402
from oracle import sbr_params as P  # noqa: E402

COMP = "V Si Ss Xi Xs Xbh Xba Xp So Sno Snh Snd Xnd Salk".split()
BEGIN, END = "<!-- scenario-gates:begin -->", "<!-- scenario-gates:end -->"


def spectral_radius(lib, p, x, kla, ec):
    """max |eig| of d(rhs)/dx at x (the volume row/column excluded: dV/dt = ec is constant)."""
    n = 13
    J = np.empty((n, n))
    fp, fm = np.empty(14), np.empty(14)
    for j in range(n):
        d = 1e-6 * max(1.0, abs(x[j + 1]))
        xp, xm = x.copy(), x.copy()
        xp[j + 1] += d
        xm[j + 1] -= d
        lib.sbro_rhs_reaction(C.byref(p), O._p(xp), kla, ec, O._p(fp))
        lib.sbro_rhs_reaction(C.byref(p), O._p(xm), kla, ec, O._p(fm))
        J[:, j] = (fp[1:] - fm[1:]) / (2 * d)
    return float(np.abs(np.linalg.eigvals(J)).max())


def episode_figures(name, tables, lam_every=4):
    means, stds = tables
    e, t = golden("sbros_" + name), golden("sbros_%s_tight" % name)
    lib, p0 = O.lib(), O.default_params(scheme=0)
    lib.sbro_rhs_reaction.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
    nv, n = valid_calls(e), int(e["n_calls"])
    scen = int(e["scenario"])
    out = {"scenario": scen, "valid_calls": nv, "domain_exit_call": int(e["domain_exit_call"]) if "domain_exit_call" in e.files else -1}
    lam = 0.0
    for scheme in (0, 1):
        ps = O.default_params(scheme=scheme)
        worst, where = 0.0, None
        for i in range(len(e["iv_kind"])):
            if e["iv_call"][i] > nv:
                continue
            span = float(e["iv_t_end"][i] - e["iv_t_start"][i])
            x1, _ = O.reaction_interval(e["iv_x_start"][i], span, float(e["iv_Kla"][i]), float(e["iv_EC"][i]), params=ps, scheme=scheme)
            g = gate(x1, e["iv_x_end"][i])
            if g.max() > worst:
                worst, where = float(g.max()), "%s, call %d (%s)" % (COMP[int(g.argmax())], int(e["iv_call"][i]),
                                                                      "aerobic" if e["iv_kind"][i] == 1 else "anoxic")
            if scheme == 0 and i % lam_every == 0:
                lam = max(lam, spectral_radius(lib, p0, e["iv_x_start"][i].copy(), float(e["iv_Kla"][i]), float(e["iv_EC"][i])))
        b = O.OracleBatch(1, ps)
        b.reset(b.mix(means, stds, [scen], e["rnd"][None]))
        xs = []
        for k in range(n):
            b.step(e["actions"][k][None])
            xs.append(b.envs["x"][0].copy())
        xs = np.array(xs)
        m = min(valid_calls(t), nv, n - 1)
        tag = "" if scheme == 0 else "_s1"
        out.update({"open" + tag: worst, "open_where" + tag: where,
                    "closed_tight" + tag: float(gate(xs[:m], t["step_x_end"][:m]).max()),
                    "closed_default" + tag: float(gate(xs[:m], e["step_x_end"][:m]).max())})
    m = min(valid_calls(t), nv, n - 1)
    out.update({"reference_default_vs_tight": float(gate(e["step_x_end"][:m], t["step_x_end"][:m]).max()),
                "lam_max_per_day": lam, "lam_dt": lam * P.DT})
    return out


def markdown(rec):
    """One row per influent scenario (worst over its episodes); the per-episode figures are in the JSON."""
    rows = ["| influent scenario | reference-captured episodes | calls compared | open loop, worst gate of one interval: RK4 × 10 / scheme 1 | "
            "closed loop vs the reference at 1e-12, worst gate: RK4 × 10 / scheme 1 | max λ·dt (RK4 stable below 2.785) |", "|---|---|---|---|---|---|"]
    for s in sorted(rec["per_scenario"], key=int):
        eps = {n: r for n, r in rec["episodes"].items() if r["scenario"] == int(s)}
        v = rec["per_scenario"][s]
        wo = max(eps, key=lambda n: eps[n]["open"])
        short = [n for n, r in eps.items() if r["valid_calls"] < 463]
        calls = "all 463" if not short else "%s%s up to call %s (near a pole from there on)" % (
            "all 463; " if len(short) < len(eps) else "", ", ".join("`%s`" % n for n in short),
            " / ".join(str(eps[n]["valid_calls"]) for n in short))
        rows.append("| %s | %s | %s | %.2f (`%s`: %s) / %.2f | %.2f / %.2f | %.2f (%.0f d⁻¹) |" % (
            s, ", ".join("`%s`" % n for n in eps), calls, v["open"], wo, eps[wo]["open_where"], v["open_s1"], v["closed_tight"],
            v["closed_tight_s1"], v["lam_dt"], max(r["lam_max_per_day"] for r in eps.values())))
    return "\n".join(rows)


def main():
    t = golden("influent_tables")
    tables = (np.ascontiguousarray(t["means"]), np.ascontiguousarray(t["stds"]))
    rec = {"what": "RK4 (10 substeps) against the reference on every captured episode; see scripts/analysis/scenario_gates.py",
           "episodes": {}, "per_scenario": {}}
    for name in EPISODES + SCENARIO_EPISODES:
        rec["episodes"][name] = episode_figures(name, tables)
        print(name, json.dumps(rec["episodes"][name]), flush=True)
    for r in rec["episodes"].values():
        s = rec["per_scenario"].setdefault(str(r["scenario"]), {"open": 0.0, "closed_tight": 0.0, "open_s1": 0.0, "closed_tight_s1": 0.0,
                                                                   "lam_dt": 0.0})
        for k in s:
            s[k] = max(s[k], r[k])
    with open(os.path.join(ROOT, "profiles", "r05_scenario_gates.json"), "w") as f:
        json.dump(rec, f, indent=1)
    md = markdown(rec)
    print(md)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(path, "w").write(s[:a] + "\n" + md + "\n" + s[b:])


if __name__ == "__main__":
    main()
