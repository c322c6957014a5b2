"""Round 5: intervals of the per-cycle env SBR-v2 (fill / reaction / idle, fresh and carried-over cycles, random actions and
scenarios) as a SECOND population for the adaptive scheme's plan - the SBROS-v1 fixtures do not contain a fill interval that is
planned on its own, nor a carried-over start (concentrated sludge, oxygen left over from the aerated idle phase).

    python scripts/analysis/cycle_intervals.py [n_envs]     ->  /tmp/sbr_cycle_intervals.npz  (kind, x0, span, kla, loading)

Logged from oracle/sbr_cycle_ref.py in RK4 mode (the trajectory the intervals are taken from does not matter much; each interval
is then judged on its own against RK4 x 160).  Test infrastructure / analysis only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402
from oracle import sbr_cycle_ref as CR  # noqa: E402
from oracle import sbr_params as P  # noqa: E402
from oracle import sbr_ref as R  # noqa: E402


def collect(n_envs=24, seed=8, cycles=3):
    t = golden("influent_tables")
    tables = (np.ascontiguousarray(t["means"]), np.ascontiguousarray(t["stds"]))
    rs = np.random.RandomState(seed)
    log = []
    orig = CR.rk4

    def logging_rk4(f, x, t0, t1, nsub, args):
        kind = 1 if f is CR.rhs_fill else 2
        log.append((kind, np.array(x, dtype=np.float64), t1 - t0, float(args[0]),
                    np.array(args[1], dtype=np.float64) if kind == 1 else np.zeros(14)))
        return orig(f, x, t0, t1, nsub, args)
    CR.rk4 = logging_rk4
    try:
        for i in range(n_envs):
            scen = i % 8
            env = CR.SbrEnv2Ref(tables, integrator="rk4")
            env.reset(rs.randn(48), scenario=scen)
            for c in range(cycles):
                a = rs.uniform(-0.2, 1.2, 3) if i % 3 else np.array([[0.25, 0.25, 0.25], [1, 0, 0], [0, 1, 1]][(i // 3) % 3], dtype=float)
                env.step(a)
                env.influent = R.influent_mix(tables[0][scen], tables[1][scen], rs.randn(48))      # carry-over reset
                env.x0 = env.x_last.copy(); env.iv = env.x0[0]; env.qin = P.WV - env.iv
            print("env %d done: %d intervals" % (i, len(log)), flush=True)
    finally:
        CR.rk4 = orig
    return dict(kind=np.array([l[0] for l in log]), X=np.array([l[1] for l in log]).T.copy(), span=np.array([l[2] for l in log]),
                kla=np.array([l[3] for l in log]), loading=np.array([l[4] for l in log]).T.copy())


if __name__ == "__main__":
    d = collect(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
    np.savez_compressed("/tmp/sbr_cycle_intervals.npz", **d)
    print("saved", d["X"].shape)
