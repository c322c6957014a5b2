"""Does the fused rollout slow down when it runs for many milliseconds?  Twelve 463-call launches (reset between them, as
bench.py --workload config5 does), each timed by HIP events; then the same after 0.3 s of sustained stepping."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from gym_sbr2_amd import SbrOSVec, _capi
N = 65536
cfg = _capi.default_config(); cfg.act_DO_max = 2.5
env = SbrOSVec(N, config=cfg)
scen = (4 + torch.arange(N, device="cuda") % 4).to(torch.int32)
def episode(calls):
    env.reset(seed=1, scenario=scen)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(calls, 77); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for label, prime in (("cold", 0.0), ("after 0.3 s of stepping", 0.3), ("after 2 s of rollouts", 2.0)):
    t0 = time.perf_counter()
    a = torch.rand(N, 2, device="cuda")
    while time.perf_counter() - t0 < prime:
        if prime > 1: episode(463)
        else:
            env.reset(seed=1, scenario=scen)
            for _ in range(400): env.step(a)
            torch.cuda.synchronize()
    ms = [episode(463) for _ in range(12)]
    print("%-26s 463-call launches, ms each: %s" % (label, " ".join("%.2f" % m for m in ms)))
    print("%-26s 462-call launch (no terminal phases): %.2f ms = %.2f us per call" % (label, episode(462), episode(462) / 0.462))
