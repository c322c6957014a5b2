"""Per-launch time of sbr_step at N = 65536 as a function of how long the GPU has been busy (clock/power ramp)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_sbr2_amd import SbrOSVec
N = 65536
env = SbrOSVec(N)
scen = (torch.arange(N, device="cuda") % 8).to(torch.int32)
env.reset(seed=1, scenario=scen)
a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
torch.cuda.synchronize()
t_start = time.perf_counter()
busy_steps = 0
for rep in range(400):               # every block is 400 REAL steps of a fresh episode (an episode has 463 calls)
    env.reset(seed=rep, scenario=scen)
    env.timer_start()
    for _ in range(400):
        env.step(a)
    ms = env.timer_stop()
    if rep < 10 or rep % 20 == 0:
        print("t = %6.3f s  block %3d  per launch %.2f us" % (time.perf_counter() - t_start, rep, ms * 1e3 / 400), flush=True)
# idle gap, then again
time.sleep(2.0)
env.reset(seed=999, scenario=scen)
env.timer_start()
for _ in range(400): env.step(a)
print("after 2 s idle: %.2f us" % (env.timer_stop() * 1e3 / 400))
