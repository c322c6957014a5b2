"""Round 6: what would fewer, longer macro intervals in the idle phase of the done call cost in accuracy and save in steps?
The idle phase (gym_SBR_oneshot.py:2554-2597: one odeint over ~464 dt with Kla held) is cut into ceil(rows/10) = 47 macro intervals
of scheme 1, each planned like a control interval (oracle/sbr_ref.py b5a_span).  For every committed episode fixture: the idle phase
from the reference's own post-draw state with m = 47, 24, 16, 12, 8 macro intervals - total Butcher-5 steps, and the end state's
distance from the reference's (its default tolerance) and from LSODA at 1e-12, in units of the parity gate.
python scripts/analysis/idle_macro_study.py   (CPU; test infrastructure)"""
import glob
import os
import sys

import numpy as np
from scipy.integrate import odeint

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sbr_ref as R  # noqa: E402

SCALE = np.array([1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10.0])


def gate(a, ref):
    return float((np.abs(a - ref) / (1e-5 * np.abs(ref) + 1e-5 * SCALE)).max())


def idle(x, span, m, kla):
    steps, per = 0, []
    hm = span / m
    for _ in range(m):
        x, n = R.b5a_macro(2, x, hm, kla)
        steps += n; per.append(n)
    return x, steps, per


rows, ref_vs_tight = {}, []
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "sbros_*.npz")))
for f in files:
    d = np.load(f)
    if "term_x_after_draw" not in d.files or int(d["crashed"]):
        continue
    x0, kla = d["term_x_after_draw"].astype(np.float64), float(d["term_Kla_idle"])
    t0, t1, nr = float(d["term_t_idle_start"]), float(d["term_t_idle_end"]), int(d["term_n_idle_rows"])
    ref = d["term_x_after_idle"]
    tight = odeint(R.rhs_idle, x0, np.linspace(t0, t1, nr), args=(kla,), rtol=1e-12, atol=1e-12)[-1]
    ref_vs_tight.append(gate(ref, tight))
    for m in (47, 24, 16, 12, 8):
        x1, steps, per = idle(x0.copy(), t1 - t0, m, kla)
        rows.setdefault(m, []).append((steps, gate(x1, ref), gate(x1, tight), max(per), os.path.basename(f)))
print("%d episodes; the reference's own default run against LSODA 1e-12: worst %.3f of the gate" % (len(rows[47]), max(ref_vs_tight)))
for m, r in rows.items():
    st = np.array([v[0] for v in r]); g = np.array([v[1] for v in r]); gt = np.array([v[2] for v in r])
    w = r[int(np.argmax(gt))]
    print("m = %2d: steps mean %.1f max %d (largest count in one macro interval %d) | vs reference: worst %.4f | vs LSODA 1e-12: worst %.4f median %.5f (%s)"
          % (m, st.mean(), st.max(), max(v[3] for v in r), g.max(), gt.max(), np.median(gt), w[4]))
