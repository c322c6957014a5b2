"""Where the parked (WAVES = 2) build of k_step first differs from the one-wave build: same global envs, call by call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_sbr2_amd as G
n_big, lo, hi = 65536 + 320, 0, 4096
gen = torch.Generator(device="cuda"); gen.manual_seed(31)
pool = torch.rand(16, n_big, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")
pool[:, ::3, 1] = 0.0
scen = (torch.arange(n_big, device="cuda") % 8).to(torch.int32)
big = G.SbrOSVec(n_big); small = G.SbrOSVec(hi - lo, first_env_id=lo)
big.reset(seed=17, scenario=scen); small.reset(seed=17, scenario=scen[lo:hi].contiguous())
shown = 0
for c in range(463):
    a = pool[c & 15]
    big.step(a); small.step(a[lo:hi].contiguous())
    xb, cb = big.get_state(); xs, cs = small.get_state()
    dx = (xb[:, lo:hi] != xs); dc = (cb[:, lo:hi] != cs) & ~(torch.isnan(cs) & torch.isnan(cb[:, lo:hi]))
    if dx.any() or dc.any():
        rx = dx.any(dim=1).nonzero().flatten().tolist(); rc = dc.any(dim=1).nonzero().flatten().tolist()
        print("call", c, "x rows", rx, "ctrl rows", rc, "envs differing", int((dx.any(dim=0) | dc.any(dim=0)).sum()))
        for r in rc[:6]:
            j = int(dc[r].nonzero()[0]); print("   ctrl", r, "env", j, float(cb[r, lo + j]), float(cs[r, j]))
        for r in rx[:4]:
            j = int(dx[r].nonzero()[0]); print("   x", r, "env", j, float(xb[r, lo + j]), float(xs[r, j]))
        shown += 1
        if shown >= 3: break
print("done", c)
