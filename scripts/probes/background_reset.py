"""Does a k_reset of ANOTHER handle, queued on a second stream while the main handle steps, hide under the steps?
k_step (scheme 1, 65 536 envs) holds one 320-register wave per SIMD and issues during about half of its cycles; a k_reset wave
(177 registers) fits next to it.  Measures the wall time of whole episodes of the main handle (events on its stream) with and
without a concurrent reset of a second 65 536-env handle queued at call 50 of the episode; and the serial cost for comparison
(the reset queued on the SAME stream).  usage: python scripts/probes/background_reset.py [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_sbr2_amd as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
main, other = G.SbrOSVec(N), G.SbrOSVec(N)
gid = torch.arange(N, device="cuda")
scen = (4 + gid % 4).to(torch.int32)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
pool = torch.rand(64, N, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
side = torch.cuda.Stream(priority=0)
side_low = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0])          # lowest priority the device offers


def episode(mode, seed):
    main.reset(seed=seed, scenario=scen)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for c in range(463):
        if c == 50:
            if mode == "same":
                other.reset(seed=seed + 100, scenario=scen)
            elif mode in ("side", "side_low"):
                s = side if mode == "side" else side_low
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    other.reset(seed=seed + 100, scenario=scen)
        main.step(pool[c & 63])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    episode("none", 1)
for rnd in range(3):
    for mode in ("none", "same", "side", "side_low"):
        v = sorted(episode(mode, 2 + i) for i in range(5))[2]
        print("round %d %-9s episode %.1f us  (%.3f us per call)" % (rnd, mode, v, v / 463), flush=True)
