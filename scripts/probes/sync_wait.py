"""Fixed cost of a SHORT timed region (the driver's `bench.py --steps 20`): wall time of [20 step() launches + synchronise]
against the device time of the 20 kernels, under different host wait modes of the HIP runtime.  Each mode runs in its own
child process (the mode has to be chosen before the runtime starts); one child at a time.
usage: python scripts/probes/sync_wait.py            (parent)
       python scripts/probes/sync_wait.py child MODE  (one measurement)"""
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODES = {
    "default": {},
    "spin_flag": {"SBR_PROBE_SPIN": "1"},                      # hipSetDeviceFlags(hipDeviceScheduleSpin)
    "active_wait_1ms": {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"},    # us of active wait before the blocking wait
    "no_interrupt": {"HSA_ENABLE_INTERRUPT": "0"},             # the ROCr signal waits poll instead of sleeping
}


def child():
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    import torch
    from gym_sbr2_amd import SbrOSVec
    if os.environ.get("SBR_PROBE_SPIN") == "1":
        hip = C.CDLL("libamdhip64.so")          # the runtime torch has loaded
        rc = hip.hipSetDeviceFlags(C.c_uint(1))   # hipDeviceScheduleSpin
        print("hipSetDeviceFlags(spin) ->", rc, flush=True)
    N, K = 65536, 20
    env = SbrOSVec(N)
    sc = (4 + torch.arange(N, device="cuda") % 4).to(torch.int32)
    env.reset(seed=1, scenario=sc)
    pool = torch.rand(64, N, 2, device="cuda") * torch.tensor([2.5, 15.0], device="cuda")
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for j in range(100):
            env.step(pool[j & 63])
        torch.cuda.synchronize()
    walls, devs, syncs = [], [], []
    for rep in range(60):
        if rep % 20 == 0:
            env.reset(seed=1, scenario=sc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for j in range(K):
            env.step(pool[j])
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6); devs.append(e0.elapsed_time(e1) * 1e3); syncs.append((t1 - t0) * 1e6)
    q = lambda v: (statistics.median(v), min(v))
    print("wall median %.1f us (min %.1f) | device median %.1f us (min %.1f) | host issue median %.1f us | fixed cost median %.1f us"
          % (*q(walls), *q(devs), statistics.median(syncs), statistics.median([w - d for w, d in zip(walls, devs)])), flush=True)
    print("first repetitions, wall - device (us):", " ".join("%.1f" % (w - d) for w, d in list(zip(walls, devs))[:6]), flush=True)
    # the shape of bench.py's timed region: a reset and five warm-up steps just before, one repetition
    for variant in ("reset+5", "5 only", "reset+5, 2 ms idle before"):
        out = []
        for rep in range(8):
            if variant != "5 only":
                env.reset(seed=1, scenario=sc)
            for j in range(5):
                env.step(pool[j])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); torch.cuda.synchronize()
            if variant.endswith("before"):
                time.sleep(0.002)
            t0 = time.perf_counter()
            e0.record()
            for j in range(K):
                env.step(pool[5 + j])
            e1.record()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.append((t2 - t0) * 1e6 - e0.elapsed_time(e1) * 1e3)
        print("%-28s wall - device (us): %s" % (variant, " ".join("%.1f" % v for v in out)), flush=True)
    # what recording the opening event costs on an idle stream
    for what in ("fresh torch event", "re-recorded torch event", "sbr_timer_start"):
        out = []
        ev = torch.cuda.Event(enable_timing=True); ev.record()
        for rep in range(12):
            for j in range(5):
                env.step(pool[j])
            if what.startswith("fresh"):
                ev = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if what == "sbr_timer_start":
                env.timer_start()
            else:
                ev.record()
            out.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
        print("%-26s %s us" % (what, " ".join("%.1f" % v for v in out)), flush=True)
    # the RL-loop pattern: one step, then wait for it
    lat = []
    for j in range(200):
        t0 = time.perf_counter()
        env.step(pool[j & 63]); torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
    print("step + synchronise: median %.1f us (min %.1f)" % q(lat), flush=True)
    env.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for name, extra in MODES.items():
            print("==", name, extra, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env={**os.environ, **extra}, check=False, timeout=240)
