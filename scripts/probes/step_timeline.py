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
cfg = _capi.default_config()
if os.environ.get("AB_SCHEME"):                    # 0 = RK4 x substeps, 1 = adaptive Butcher-5 (the default)
    cfg.scheme = int(os.environ["AB_SCHEME"])
for N in sizes:
    env = SbrOSVec(N, config=cfg)
    scen = (torch.arange(N, device="cuda") % 8).to(torch.int32)
    a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
    if os.environ.get("SBR_TL_POLICY") == "dose":      # NO3 set-point 0: every lane doses carbon in the anoxic phases
        a[:, 1] = 0.0
    waves = (N + 63) // 64
    buf = torch.zeros(waves, 16, dtype=torch.int64, device="cuda")
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
        names = {0: "entry", 1: "loads returned", 2: "before intervals", 3: "RK4 done", 4: "reward done", 8: "window rolled",
                 9: "plant stores issued", 10: "controller stores issued", 5: "state stores issued (+trace)", 11: "reward/done stores issued",
                 12: "output values formed", 6: "output stores issued", 7: "stores acknowledged"}
        order = [k for k in (0, 1, 2, 3, 4, 8, 9, 10, 5, 11, 12, 6, 7) if t[:, k].max() > 0]
        for k in order:
            print("   %2d %-30s %7.2f %7.2f %7.2f" % (k, names[k], rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
        print("   per-wave segment medians: " + "  ".join("%d->%d %.2f" % (a, b_, np.median(rel[:, b_] - rel[:, a])) for a, b_ in zip(order[:-1], order[1:])))
    env.close()
