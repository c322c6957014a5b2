"""Round 6: how often does the device's plan (SBR_C_PLAN) differ from the oracle's when both start a call from the same state?
Lockstep over whole episodes of 4096 envs (the workload of test_config2_4096_envs_full_episode_against_oracle: even envs uniform
set-points, odd envs U[0, 2.5] for the DO set-point), several seeds; counts mismatches among envs that are well-posed before and after the
call, and among the rest.   python scripts/debug/plan_flip_soak.py [first_seed] [n_seeds]   (GPU box; test infrastructure)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import SbrOSVec, _capi  # noqa: E402
from gym_sbr2_amd.vec_env import load_influent_tables  # noqa: E402
from oracle import sbr_oracle as O  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
means, stds = load_influent_tables()
n, ncall = 4096, 463
tot_ok = tot_rest = flips_ok = flips_rest = 0
worst = 0.0
scale = np.array([1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10.0])
for seed in range(first, first + count):
    rs = np.random.RandomState(seed)
    scen = (np.arange(n) % 8).astype(np.int32)
    rnd = rs.randn(n, 48)
    env = SbrOSVec(n, out_dtype=torch.float32)
    sync = O.OracleBatch(n, O.default_params(scheme=1), nthreads=16)
    env.reset(scenario=scen, rnd=rnd)
    sync.reset(sync.mix(means, stds, scen, rnd))
    x, ctrl = env.get_state()
    odd = np.arange(n) % 2 == 1
    for c in range(ncall):
        a = np.column_stack([rs.uniform(0, 8, n), rs.uniform(0, 15, n)])
        a[odd, 0] = rs.uniform(0, 2.5, odd.sum())
        a = a.astype(np.float32)
        xb, cb = x.cpu().numpy(), ctrl.cpu().numpy()
        sync.load_state(xb, cb)
        pole_before = (cb[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) != 0
        env.step(torch.from_numpy(a).cuda())
        sync.step(a.astype(np.float64))
        x, ctrl = env.get_state()
        cn = ctrl.cpu().numpy()
        ok = ~pole_before & ((cn[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) == 0)
        dev, ref = cn[_capi.C_PLAN].astype(np.int64), sync.envs["scheme_plan"].astype(np.int64) & 0xff
        diff = dev != ref
        flips_ok += int((diff & ok).sum()); flips_rest += int((diff & ~ok).sum())
        tot_ok += int(ok.sum()); tot_rest += int((~ok).sum())
        g = np.abs(x.cpu().numpy().T - sync.envs["x"]) / (1e-5 * np.abs(sync.envs["x"]) + 1e-5 * scale)
        worst = max(worst, float(g[ok].max()))
    env.close()
    print("seed %d done: %d well-posed env-calls so far, %d plan mismatches among them; %d / %d among the env-calls near a pole; worst lockstep gate %.2e"
          % (seed, tot_ok, flips_ok, flips_rest, tot_rest, worst), flush=True)
