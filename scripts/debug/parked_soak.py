"""Soak of the two-waves build of k_step against the one-wave build: E episodes of a 262144-env handle (float64 outputs and
actions every other episode) against a 65536-env handle on global ids 98304..163839, uniform policy with a third of the lanes
dosing; plant, controller rows, returns and the done call's outputs must agree bit for bit after every episode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_sbr2_amd as G

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_big, lo, hi = 262144, 98304, 163840
bad = 0
for ep in range(E):
    f64 = ep % 2 == 1
    kw = dict(out_dtype=torch.float64, action_dtype=torch.float64) if f64 else {}
    big, one = G.SbrOSVec(n_big, **kw), G.SbrOSVec(hi - lo, first_env_id=lo, **kw)
    gen = torch.Generator(device="cuda"); gen.manual_seed(100 + ep)
    dt = torch.float64 if f64 else torch.float32
    pool = (torch.rand(8, n_big, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")).to(dt)
    pool[:, ep % 3::3, 1] = 0.0
    scen = ((torch.arange(n_big, device="cuda") + ep) % 8).to(torch.int32)
    ob = big.reset(seed=7 + ep, scenario=scen)
    ok = torch.equal(one.reset(seed=7 + ep, scenario=scen[lo:hi].contiguous()), ob[lo:hi])
    for c in range(463):
        a = pool[c & 7]
        o, s, r, d = big.step(a)
        o2, s2, r2, d2 = one.step(a[lo:hi].contiguous())
    ok = ok and torch.equal(o2, o[lo:hi]) and torch.equal(s2, s[lo:hi]) and torch.equal(r2, r[lo:hi]) and bool(d.all()) and bool(d2.all())
    xb, cb = big.get_state(); x1, c1 = one.get_state()
    same = lambda u, v: bool(((u == v) | (torch.isnan(u) & torch.isnan(v))).all())      # noqa: E731
    ok = ok and same(x1, xb[:, lo:hi]) and same(c1, cb[:, lo:hi]) and torch.equal(one.episode_returns(), big.episode_returns()[lo:hi])
    st = big.status() if hasattr(big, "status") else None
    print("episode %d (%s): %s" % (ep, "f64" if f64 else "f32", "identical" if ok else "DIFFERENT"), flush=True)
    bad += 0 if ok else 1
    big.close(); one.close()
print("soak:", "ok" if bad == 0 else "%d episodes differ" % bad)
sys.exit(1 if bad else 0)
