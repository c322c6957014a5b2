"""Where does a k_step launch spend its time?  T(substeps) = fixed + substeps * per_substep, and T(N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_sbr2_amd import SbrOSVec, _capi

def time_steps(N, substeps, K=300, outputs=True):
    cfg = _capi.default_config(); cfg.substeps = substeps
    env = SbrOSVec(N, config=cfg)
    env.reset(seed=1, scenario=(torch.arange(N, device="cuda") % 8).to(torch.int32))
    a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
    if not outputs:
        import ctypes as C
        step = lambda: _capi.check(env.lib.sbr_step(env._h, C.c_void_p(a.data_ptr()), None, None, None, None, env._stream()), env._h)
    else:
        step = lambda: env.step(a)
    for _ in range(30): step()
    torch.cuda.synchronize(); env.timer_start()
    for _ in range(K): step()
    ms = env.timer_stop(); env.close()
    return ms * 1e3 / K

print("device", torch.cuda.get_device_name(0))
for N in (65536,):
    ts = {s: time_steps(N, s) for s in (1, 2, 5, 10, 20)}
    slope = (ts[20] - ts[1]) / 19
    print("N=%d  us/launch by substeps: %s  => per substep %.3f us, fixed %.2f us" % (N, {k: round(v, 2) for k, v in ts.items()}, slope, ts[1] - slope))
    print("   without obs/state/reward/done outputs, substeps=10: %.2f us ; substeps=1: %.2f us" % (time_steps(N, 10, outputs=False), time_steps(N, 1, outputs=False)))
for N in (4096, 16384, 32768, 65536, 98304, 131072, 262144, 524288):
    t = time_steps(N, 10, K=200)
    print("N=%7d (%.2f waves/SIMD): %.2f us/launch  %.3e env-steps/s  %.1f GB/s algorithmic" % (N, N / 65536, t, N / t * 1e6, N * 513 / t / 1e3))
