"""A/B timing of kernel variants (build/libsbr_amd_<name>.so): each variant in its own subprocess, two rounds,
per-launch device time of sbr_step at several batch sizes.  usage: python scripts/gpu_ab.py name1 name2 ... [-- N1 N2 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from gym_sbr2_amd import SbrOSVec
out = []
for N in %r:
    env = SbrOSVec(N)
    env.reset(seed=1, scenario=(torch.arange(N, device="cuda") %% 8).to(torch.int32))
    a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
    import time as _t
    t0 = _t.perf_counter()
    while _t.perf_counter() - t0 < 0.2:          # steady clocks: the GPU needs ~25 ms of sustained work (probes/clock_ramp.py)
        env.reset(seed=1, scenario=(torch.arange(N, device="cuda") %% 8).to(torch.int32))
        for _ in range(400): env.step(a)
        torch.cuda.synchronize()
    env.reset(seed=1, scenario=(torch.arange(N, device="cuda") %% 8).to(torch.int32))
    for _ in range(40): env.step(a)
    torch.cuda.synchronize(); env.timer_start()
    for _ in range(300): env.step(a)
    t = env.timer_stop() * 1e3 / 300
    env.reset(seed=1, scenario=(torch.arange(N, device="cuda") %% 8).to(torch.int32)); torch.cuda.synchronize()
    env.timer_start(); env.rollout(462, 3); r = env.timer_stop() * 1e3 / 462
    out.append("%%d: step %%.2f us (%%.2fe9/s) rollout %%.2f us/call (%%.2fe9/s)" %% (N, t, N / t / 1e3, r, N / r / 1e3))
    env.close()
print(" | ".join(out))
'''
args = sys.argv[1:]
names = args[:args.index("--")] if "--" in args else args
sizes = [int(v) for v in args[args.index("--") + 1:]] if "--" in args else [65536, 131072, 262144]
for rnd in range(2):
    for name in names:
        env = dict(os.environ, SBR_AMD_LIB=os.path.join(ROOT, "build", "libsbr_amd_%s.so" % name))
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, sizes)], env=env, capture_output=True, text=True)
        print("round %d %-10s %s" % (rnd, name, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
