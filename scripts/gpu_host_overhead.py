"""How fast can the host issue sbr_step launches?  (a) SbrOSVec.step, (b) raw ctypes call with prebuilt arguments,
(c) a captured graph of 64 steps replayed.  Kernel time at N=65536 is ~24 us; the host must stay below that."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gym_sbr2_amd import SbrOSVec, _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = SbrOSVec(N)
scen = (torch.arange(N, device="cuda") % 8).to(torch.int32)
env.reset(seed=1, scenario=scen)
pool = torch.rand(64, N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
K = 448
def timed(fn, label, per=1):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s host issue %.2f us/step, wall incl. GPU %.2f us/step" % (label, (t1 - t0) * 1e6 / K, (t2 - t0) * 1e6 / K), flush=True)
def a():
    for j in range(K): env.step(pool[j & 63])
timed(a, "(a) SbrOSVec.step"); env.reset(seed=1, scenario=scen)
lib, h = env.lib, env._h
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
args = [(C.c_void_p(pool[j].data_ptr()), C.c_void_p(env.obs.data_ptr()), C.c_void_p(env.state.data_ptr()),
         C.c_void_p(env.reward.data_ptr()), C.c_void_p(env.done.data_ptr())) for j in range(64)]
def b():
    f = lib.sbr_step
    for j in range(K):
        a_, o, s, r, d = args[j & 63]
        f(h, a_, o, s, r, d, st)
timed(b, "(b) raw ctypes, prebuilt arguments"); env.reset(seed=1, scenario=scen)
# (c) graph of 64 steps
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for j in range(3): env.step(pool[j])          # warm up on the capture stream
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for j in range(64): env.step(pool[j])
torch.cuda.current_stream().wait_stream(side)
env.reset(seed=1, scenario=scen); torch.cuda.synchronize()
def c():
    for _ in range(K // 64): g.replay()
timed(c, "(c) torch CUDAGraph of 64 steps, replayed")
x, ctrl = env.get_state(); print("steps row after graph replays:", ctrl[_capi.C_STEPS][:3].tolist(), "finite:", bool(torch.isfinite(x).all()))
env.close()
