"""Timing of the per-cycle kernel k_cycle (SBR-v2): one launch = one whole cycle = 528 control intervals per env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gym_sbr2_amd import SbrEnv2Vec
for N in (4096, 65536, 262144):
    env = SbrEnv2Vec(N)
    scen = (torch.arange(N, device="cuda") % 8).to(torch.int32)
    a = torch.rand(N, 3, device="cuda")
    env.reset(seed=1, scenario=scen); env.step(a); torch.cuda.synchronize()
    K = 5
    env.timer_start()
    for _ in range(K):
        env.reset(seed=2, scenario=scen); env.step(a)
    ms = env.timer_stop() / K
    print("N=%7d: %.2f ms per reset+cycle -> %.3e cycles/s = %.3e control intervals/s (528 per cycle); reward mean %.4f" % (
        N, ms, N / ms * 1e3, N * 528 / ms * 1e3, float(env.reward.mean())))
    env.close()
