"""Device time of the done call (call 463 of an episode: last interval + settle + draw + idle phase) of sbr_step, by HIP events, for
the library selected by SBR_AMD_LIB (default: in-tree).  usage: python scripts/probes/done_call.py [n_envs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import SbrOSVec  # noqa: E402

uniform = "--uniform" in sys.argv          # SURVEY 8d's inputs: U[0, 8] x U[0, 15] on all eight scenarios (86 % of the envs leave the model's domain)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 65536
env = SbrOSVec(n)
gid = torch.arange(n, device="cuda")
scen = ((gid % 8) if uniform else (4 + gid % 4)).to(torch.int32)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
pool = torch.rand(64, n, 2, device="cuda", generator=gen) * torch.tensor([8.0 if uniform else 2.5, 15.0], device="cuda")
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
last, rest = [], []
marks = {c: torch.cuda.Event(enable_timing=True) for c in (99, 199, 299, 399)}
for ep in range(6):
    env.reset(seed=1000 + ep, scenario=scen)
    torch.cuda.synchronize()
    e0.record()
    for c in range(462):
        env.step(pool[c & 63])
        if c in marks:
            marks[c].record()
    e1.record()
    env.step(pool[462 & 63])
    e2.record()
    torch.cuda.synchronize()
    if ep:
        rest.append(e0.elapsed_time(e1) * 1e3 / 462); last.append(e1.elapsed_time(e2) * 1e3)
torch.cuda.synchronize()
seg = [e0.elapsed_time(marks[99]) * 10, marks[99].elapsed_time(marks[199]) * 10, marks[199].elapsed_time(marks[299]) * 10,
       marks[299].elapsed_time(marks[399]) * 10, marks[399].elapsed_time(e1) * 1e3 / 62]
print("   last episode, us per call over calls 0-99 / 100-199 / 200-299 / 300-399 / 400-461: " + " / ".join("%.2f" % v for v in seg))
print("%s%s: calls 1..462 %.2f us each (eager issue), the done call %.1f us (five episodes: %s)"
      % (os.environ.get("SBR_AMD_LIB", "in-tree"), " (uniform policy)" if uniform else "", sum(rest) / len(rest), sum(last) / len(last), ", ".join("%.1f" % v for v in last)))
env.close()
