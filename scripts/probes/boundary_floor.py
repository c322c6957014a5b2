"""What a kernel boundary costs with k_step's own footprint (1.1 KB of arguments, 57 KB of LDS per workgroup, 250 VGPRs, 256
workgroups): the diagnostic build (-DSBR_STAMPS) returns at the top of k_step when the stamp pointer is 1, so the
launch-to-launch period of that empty kernel is everything a launch costs outside the waves' work.
usage: SBR_AMD_LIB=build/libsbr_amd_stamps.so python scripts/probes/boundary_floor.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_sbr2_amd import SbrOSVec, _capi

lib = _capi.load()
lib.sbr_set_stamps.restype = C.c_int
lib.sbr_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for N in (4096, 65536, 262144):
    env = SbrOSVec(N)
    env.reset(seed=1, scenario=(torch.arange(N, device="cuda") % 8).to(torch.int32))
    a = torch.rand(N, 2, device="cuda") * torch.tensor([2.5, 15.0], device="cuda")
    out = []
    for empty in (False, True):
        lib.sbr_set_stamps(env._h, C.c_void_p(1 if empty else 0))
        for _ in range(3):
            for _ in range(400):
                env.step(a)
            torch.cuda.synchronize()
        env.reset(seed=1, scenario=(torch.arange(N, device="cuda") % 8).to(torch.int32))
        for _ in range(40):
            env.step(a)
        torch.cuda.synchronize(); env.timer_start()
        for _ in range(300):
            env.step(a)
        out.append(env.timer_stop() * 1e3 / 300)
        # the same through a captured HIP graph of 64 launches (no host work between launches)
        g = env.capture_steps([a] * 64)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize(); env.timer_start()
        for _ in range(10):
            g.replay()
        out.append(env.timer_stop() * 1e3 / 640)
        del g
    print("N = %6d: k_step %.2f us per launch eager, %.2f us in a graph; the same kernel returning at once: %.2f us eager "
          "(host-bound: Python + launch), %.2f us in a graph" % (N, out[0], out[1], out[2], out[3]))
    env.close()
