"""Digest of what a build of the library computes, for comparing two builds bit for bit (kernel A/B experiments: run once with
SBR_AMD_LIB set and once without, the lines must be equal).  Episodes of the uniform policy with a third of the lanes dosing at
4096, 65536 and 262144 envs (the three launch shapes of k_step), float32 and float64 outputs; sha256 over every call's outputs,
the plant, the controller rows and the returns.   python scripts/debug/lib_digest.py   (GPU box; test infrastructure)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_sbr2_amd as G

for n, f64 in ((4096, False), (4096, True), (65536, False), (262144, False), (262144, True)):
    kw = dict(out_dtype=torch.float64, action_dtype=torch.float64) if f64 else {}
    env = G.SbrOSVec(n, **kw)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5 + n)
    pool = (torch.rand(8, n, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")).to(torch.float64 if f64 else torch.float32)
    pool[:, 1::3, 1] = 0.0
    scen = (torch.arange(n, device="cuda") % 8).to(torch.int32)
    h = hashlib.sha256()
    h.update(env.reset(seed=3, scenario=scen).cpu().numpy().tobytes())
    for c in range(463):
        o, s, r, d = env.step(pool[c & 7])
        if c % 7 == 0 or c > 455:
            for t in (o, s, r, d):
                h.update(t.cpu().numpy().tobytes())
    x, ctrl = env.get_state()
    for t in (x, ctrl, env.episode_returns()):
        h.update(t.cpu().numpy().tobytes())
    print("%d envs %s: %s" % (n, "f64" if f64 else "f32", h.hexdigest()[:24]), flush=True)
    env.close()
