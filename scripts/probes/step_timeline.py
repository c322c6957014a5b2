"""Where the time of one k_step launch goes: a diagnostic build of the library (-DSBR_STAMPS, build/libsbr_amd_stamps.so)
records the 100 MHz real-time counter at eight points of the kernel for every wave; this script runs a few hundred steps
at steady clocks, stamps ONE launch in the middle of a back-to-back sequence, and prints per stamp the offset from the
earliest wave's entry (min / median / max over waves) in microseconds.

    0 entry   1 loads returned + parked   2 before the intervals   3 PIDs + RK4 done   4 reward done
    5 state stores issued   6 output stores issued   7 all stores acknowledged

usage: SBR_AMD_LIB=build/libsbr_amd_stamps.so python scripts/probes/step_timeline.py [N ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gym_sbr2_amd import SbrOSVec, _capi

sizes = [int(v) for v in sys.argv[1:]] or [65536, 4096]
lib = _capi.load()
lib.sbr_set_stamps.restype = C.c_int
lib.sbr_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for N in sizes:
    env = SbrOSVec(N)
    scen = (torch.arange(N, device="cuda") % 8).to(torch.int32)
    a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
    if os.environ.get("SBR_TL_POLICY") == "dose":      # NO3 set-point 0: every lane doses carbon in the anoxic phases
        a[:, 1] = 0.0
    waves = (N + 63) // 64
    buf = torch.zeros(waves, 8, dtype=torch.int64, device="cuda")
    for phase_calls, label in ((20, "anoxic (dosing code path)"), (120, "aerobic (no dosing)")):
        env.reset(seed=1, scenario=scen)
        for _ in range(3):                      # steady clocks
            env.reset(seed=1, scenario=scen)
            for _ in range(400):
                env.step(a)
            torch.cuda.synchronize()
        env.reset(seed=1, scenario=scen)
        for _ in range(phase_calls):
            env.step(a)
        lib.sbr_set_stamps(env._h, C.c_void_p(buf.data_ptr()))
        env.step(a)
        lib.sbr_set_stamps(env._h, None)
        for _ in range(5):
            env.step(a)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.float64)
        t0 = t[:, 0].min()
        rel = (t - t0) * 0.01                   # 100 MHz ticks -> us
        print("N = %d, %s: offsets from the first wave's entry, us (min / median / max over %d waves)" % (N, label, waves))
        for k, name in enumerate(["entry", "loads returned", "before intervals", "RK4 done", "reward done", "state stores issued",
                                  "output stores issued", "stores acknowledged"]):
            print("   %d %-22s %7.2f %7.2f %7.2f" % (k, name, rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
        d = np.diff(rel, axis=1)
        print("   per-wave segment medians: " + "  ".join("%d->%d %.2f" % (k, k + 1, np.median(d[:, k])) for k in range(7)))
    env.close()
