"""A/B timing of kernel variants (build/libsbr_amd_<name>.so) BY PHASE of the episode, each variant in its own subprocess,
two rounds, HIP events on the launch stream, bench.py's workload (physical policy, per-call random set-points):
  dosing   calls 5..44 of an episode (anoxic phase, the NO3-PID doses carbon: what the driver's --steps 20 --warmup 5 times)
  aerobic  calls 60..220 (closed reactor)
  episode  all 463 calls incl. the terminal one, + the reset
Under bench.py's random NO3 set-points hardly any lane doses after the first ten calls of an episode (the PID's output is
clamped at 0), so "dosing" then mostly times the closed-reactor loop; AB_POLICY=dose sets the NO3 set-point to 0, which makes
every lane dose in the anoxic phases (the reference's own anchor episode, constant action [2, 5], doses in most of them too).
AB_SCHEME=0 / 1 selects cfg.scheme (round 5).
usage: [AB_POLICY=dose] [AB_SCHEME=0] python scripts/gpu_ab2.py name1 name2 ... [-- N1 N2 ...]      (name "tree" = the in-tree library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from gym_sbr2_amd import SbrOSVec, _capi
out = []
cfg = _capi.default_config()
if os.environ.get("AB_SCHEME"):                    # 0 = RK4 x substeps, 1 = adaptive Butcher-5 (the default)
    cfg.scheme = int(os.environ["AB_SCHEME"])
for N in %r:
    env = SbrOSVec(N, config=cfg)
    gid = torch.arange(N, device="cuda")
    scen = (4 + gid %% 4).to(torch.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    pool = torch.rand(64, N, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
    if os.environ.get("AB_POLICY") == "dose":      # NO3 set-point 0: every lane doses carbon in the anoxic phases (EC saturates)
        pool[:, :, 1] = 0.0
    def run(a, b):
        for j in range(a, b): env.step(pool[j & 63])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:          # steady clocks
        env.reset(seed=1, scenario=scen); run(0, 463); torch.cuda.synchronize()
    res = {"dosing": [], "aerobic": [], "episode": []}
    for rep in range(5):
        env.reset(seed=2 + rep, scenario=scen); run(0, 5)
        torch.cuda.synchronize(); env.timer_start(); run(5, 45); res["dosing"].append(env.timer_stop() * 1e3 / 40)
        run(45, 60)
        torch.cuda.synchronize(); env.timer_start(); run(60, 220); res["aerobic"].append(env.timer_stop() * 1e3 / 160)
        run(220, 463)
        torch.cuda.synchronize(); env.timer_start(); env.reset(seed=50 + rep, scenario=scen); run(0, 463); res["episode"].append(env.timer_stop() * 1e3 / 463)
    med = lambda v: sorted(v)[len(v) // 2]
    out.append("%%d: dosing %%.2f aerobic %%.2f episode %%.2f us/call" %% (N, med(res["dosing"]), med(res["aerobic"]), med(res["episode"])))
    env.close()
print(" | ".join(out))
'''
args = sys.argv[1:]
names = args[:args.index("--")] if "--" in args else args
sizes = [int(v) for v in args[args.index("--") + 1:]] if "--" in args else [65536]
for rnd in range(2):
    for name in names:
        env = dict(os.environ)
        if name != "tree":
            env["SBR_AMD_LIB"] = os.path.join(ROOT, "build", "libsbr_amd_%s.so" % name)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, sizes)], env=env, capture_output=True, text=True)
        print("round %d %-12s %s" % (rnd, name, r.stdout.strip() or r.stderr.strip()[-400:]), flush=True)
