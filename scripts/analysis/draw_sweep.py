"""Round 6, VERDICT r5 item 4: find an env whose reactor volume after the done call is >= WV (the property
tests/test_gpu_parity.py::test_size_independent_properties_at_65536 asserted for every env, saw fail once in ten runs on one env of
65 536 and then narrowed to envs whose NEAR_POLE flag is clear) - on the CPU oracle, with the test's workload: stochastic influent
(Philox normals, scenario = id mod 8), uniform float32 set-points U[0, 8] x U[0, 15] per call, 463 calls.  Test infrastructure.

    python scripts/analysis/draw_sweep.py [--envs 10000000] [--batch 16384] [--scheme 1] [--out profiles/r06_draw_sweep.json]

For every offender the state BEFORE the done call is kept and the draw (Sim_Settling_Drawing, gym_SBR_oneshot.py:2327-2393) is
replayed in Python with its intermediates: m, the layer sludge sX, the wastage loop's `waste`, the partially wasted layer and the
quotient qw = waste / (sX[part] - biomass_setpoint)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd.vec_env import load_influent_tables  # noqa: E402
from oracle import sbr_oracle as O  # noqa: E402


def draw_replay(p, x):
    """The settle + draw of oracle/sbr_oracle.c terminal() (the same operations), with every intermediate returned."""
    xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7])
    vs = x[0]
    z = vs / p.settler_area
    t_set = p.t_settle * p.t_cycle
    a = p.settler_vmax / z * t_set
    ea = math.exp(-a)
    sx = [0.0] * 10
    term, partial = 1.0, 0.0
    for j in range(9):
        partial += term
        sx[9 - j] = xf * ea * partial
        term *= a / (j + 1)
    sx[0] = 10.0 * xf - sum(sx[1:])
    layer_v = vs / 10
    resid_v = vs - p.Qeff
    m = int(math.ceil(round(p.Qeff / layer_v)))
    m = min(max(m, 1), 9)
    w = [layer_v * sx[i] for i in range(10 - m)]
    waste0 = sum(w) - p.biomass_setpoint * resid_v
    waste, part, qw, removed = waste0, None, float("nan"), 0
    for i in range(10 - m):
        rest = waste - w[i]
        if rest > 0:
            waste = rest
            resid_v -= layer_v
            removed += 1
        else:
            part = i
            qw = waste / (sx[i] - p.biomass_setpoint)
            resid_v -= qw
            break
    return {"Xf": xf, "V_before": vs, "a": a, "m": m, "sX": sx, "layer_v": layer_v, "waste_initial": waste0, "waste_at_part": waste,
            "whole_layers_removed": removed, "part_layer": part, "sX_part_minus_setpoint": (sx[part] - p.biomass_setpoint) if part is not None else None,
            "qw": qw, "V_after": resid_v}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--scheme", type=int, default=1)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--max-found", type=int, default=24)
    ap.add_argument("--seed0", type=int, default=6000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_draw_sweep.json"))
    args = ap.parse_args()
    means, stds = load_influent_tables()
    p = O.default_params(scheme=args.scheme)
    n = args.batch
    scen = (np.arange(n) % 8).astype(np.int32)
    found, done_envs, t0 = [], 0, time.time()
    flagged_total, negq_total, v_max, qw_min, qw_max = 0, 0, -1e300, 1e300, -1e300
    k = 0
    while done_envs < args.envs and len(found) < args.max_found:
        seed = args.seed0 + k
        b = O.OracleBatch(n, p, nthreads=args.threads, first_env_id=k * n)
        b.reset(b.mix(means, stds, scen, b.normals(seed)))
        rs = np.random.RandomState(seed)
        for c in range(463):
            a = np.column_stack([rs.uniform(0, 8, n), rs.uniform(0, 15, n)]).astype(np.float32).astype(np.float64)
            if c == 462:
                before = b.envs.copy()
            _, _, _, d = b.step(a, want_obs=False)
        assert d.all()
        v = b.envs["x"][:, 0]
        st = b.envs["status"].astype(int)
        flagged_total += int(((st & 2) != 0).sum())
        negq_total += int((b.envs["qw"] < 0).sum())
        v_max = max(v_max, float(np.nanmax(v))); qw_min = min(qw_min, float(np.nanmin(b.envs["qw"]))); qw_max = max(qw_max, float(np.nanmax(b.envs["qw"])))
        bad = np.nonzero(~(v < p.WV))[0]
        for i in bad:
            # the state the draw sees is the one AFTER the last interval: replay that interval through the oracle itself
            one = O.OracleBatch(1, p, nthreads=1)
            one.envs[0] = before[i]
            p2 = O.default_params(scheme=args.scheme); p2.terminal = 0
            one.p = p2
            one.step(a[i:i + 1], want_obs=False)
            rep = draw_replay(p, [float(q) for q in one.envs["x"][0]])
            found.append({"batch_seed": seed, "env": int(i), "global_env_id": int(k * n + i), "scenario": int(scen[i]),
                          "status_bits_after_episode": int(st[i]), "status_bits_before_done_call": int(before["status"][i]),
                          "near_pole": bool(st[i] & 2), "negative": bool(st[i] & 1),
                          "x_before_settle": [float(q) for q in one.envs["x"][0]],
                          "V_after_done_call": float(v[i]), "qw_oracle": float(b.envs["qw"][i]), "draw": rep})
        done_envs += n
        k += 1
        if k % 10 == 0 or bad.size:
            print("%9d envs, %.0f s: %d with V >= WV after the done call (%d flagged NEAR_POLE so far, %d with Qw < 0; max V %.4f, Qw in [%.4g, %.4g])"
                  % (done_envs, time.time() - t0, len(found), flagged_total, negq_total, v_max, qw_min, qw_max), flush=True)
    out = {"what": "CPU-oracle sweep of test_size_independent_properties_at_65536's workload for V >= WV after the done call",
           "scheme": args.scheme, "envs": done_envs, "policy": "uniform float32 U[0, 8] x U[0, 15] per call, scenario = id mod 8, Philox influent noise",
           "near_pole_envs": flagged_total, "envs_with_negative_qw": negq_total, "max_V_after_done_call": v_max,
           "min_Qw": qw_min, "max_Qw": qw_max, "instances": found, "seconds": time.time() - t0}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out, "-", len(found), "instances in", done_envs, "envs")


if __name__ == "__main__":
    main()
