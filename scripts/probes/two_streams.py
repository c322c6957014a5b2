"""Two (or four) independent half-batches on their own streams, stepped alternately from one host thread: does the memory
phase of one overlap the arithmetic of the other?  Total envs = 65536 in every case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_sbr2_amd import SbrOSVec

def run(groups, total=65536, steps=400, reps=6):
    n = total // groups
    envs, streams, acts, scen = [], [], [], []
    for g in range(groups):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            e = SbrOSVec(n, first_env_id=g * n)
            sc = ((torch.arange(n, device="cuda") + g * n) % 8).to(torch.int32)
            a = torch.rand(n, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
        envs.append(e); streams.append(st); acts.append(a); scen.append(sc)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(reps):
        for g in range(groups):
            with torch.cuda.stream(streams[g]):
                envs[g].reset(seed=rep, scenario=scen[g])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            for g in range(groups):
                with torch.cuda.stream(streams[g]):
                    envs[g].step(acts[g])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep >= 2: best = min(best, dt)
    print("%d group(s) of %6d envs: %.2f us per step of all 65536 envs = %.2fe9 env-steps/s" % (groups, n, best / steps * 1e6, total * steps / best / 1e9), flush=True)
    for e in envs: e.close()

for g in (1, 2, 4, 1):
    run(g)
