"""Device time of the done call (call 463 of an episode: last interval + settle + draw + idle phase) of sbr_step, by HIP events, for
the library selected by SBR_AMD_LIB (default: in-tree).  usage: python scripts/probes/done_call.py [n_envs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import SbrOSVec  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = SbrOSVec(n)
gid = torch.arange(n, device="cuda")
scen = (4 + gid % 4).to(torch.int32)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
pool = torch.rand(64, n, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
last, rest = [], []
for ep in range(6):
    env.reset(seed=1000 + ep, scenario=scen)
    torch.cuda.synchronize()
    e0.record()
    for c in range(462):
        env.step(pool[c & 63])
    e1.record()
    env.step(pool[462 & 63])
    e2.record()
    torch.cuda.synchronize()
    if ep:
        rest.append(e0.elapsed_time(e1) * 1e3 / 462); last.append(e1.elapsed_time(e2) * 1e3)
print("%s: calls 1..462 %.2f us each (eager issue), the done call %.1f us (five episodes: %s)"
      % (os.environ.get("SBR_AMD_LIB", "in-tree"), sum(rest) / len(rest), sum(last) / len(last), ", ".join("%.1f" % v for v in last)))
env.close()
