"""What the output phase of k_step costs: the ABI takes NULL for every output of sbr_step, so the same library is timed with
the output sets switched off one by one (HIP events on the launch stream, back-to-back calls of the aerobic phase, calls
60..220 of an episode, bench.py's random set-points):

    all      obs + state + reward + done          (what bench.py and a training loop run)
    no_rows  reward + done only                   (no LDS transpose, no observation / state rows)
    none     every output NULL                    (plant + controller rows only)

usage: python scripts/probes/step_parts.py [N ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gym_sbr2_amd import SbrOSVec

sizes = [int(v) for v in sys.argv[1:]] or [65536, 4096]
for N in sizes:
    env = SbrOSVec(N)
    lib, h = env.lib, env._h
    scen = (4 + torch.arange(N, device="cuda") % 4).to(torch.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    pool = torch.rand(64, N, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
    obs = torch.empty(N, 18, device="cuda"); state = torch.empty(N, 15, device="cuda")
    rew = torch.empty(N, device="cuda"); done = torch.empty(N, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    modes = {"all": (obs.data_ptr(), state.data_ptr(), rew.data_ptr(), done.data_ptr()),
             "no_rows": (None, None, rew.data_ptr(), done.data_ptr()),
             "none": (None, None, None, None)}

    def run(a, b, outs):
        for j in range(a, b):
            rc = lib.sbr_step(h, C.c_void_p(pool[j & 63].data_ptr()), outs[0], outs[1], outs[2], outs[3], st)
            assert rc == 0

    for _ in range(3):                          # steady clocks
        env.reset(seed=1, scenario=scen); run(0, 463, modes["all"]); torch.cuda.synchronize()
    res = {k: [] for k in modes}
    for rep in range(4):
        for k, outs in modes.items():
            env.reset(seed=2 + rep, scenario=scen); run(0, 60, outs)
            torch.cuda.synchronize(); env.timer_start(); run(60, 220, outs); res[k].append(env.timer_stop() * 1e3 / 160)
    print("N = %6d  " % N + "  ".join("%s %.2f (min %.2f)" % (k, sorted(v)[len(v) // 2], min(v)) for k, v in res.items()) + "  us per call")
    env.close()
