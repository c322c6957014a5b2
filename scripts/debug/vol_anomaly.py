"""Debug (round 5): envs whose volume is not below WV after the done call under the uniform policy."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gym_sbr2_amd as G
from gym_sbr2_amd import _capi
n = 65536
env = G.SbrOSVec(n)
scen = (np.arange(n) % 8).astype(np.int32)
env.reset(seed=5, scenario=scen)
torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
acts = torch.rand(463, n, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
prev = None
for c in range(463):
    if c == 462:
        prev = [t.clone() for t in env.get_state()]
    o, s, r, d = env.step(acts[c])
x, ctrl = env.get_state()
bad = torch.nonzero(~(x[0] < 1.32)).flatten().cpu().numpy()
print("bad envs:", len(bad), bad[:10])
st = ctrl[_capi.C_STATUS].cpu().numpy().astype(int)
print("status of bad:", st[bad[:10]])
for i in bad[:4]:
    print("env", i, "x before done call", prev[0][:, i].cpu().numpy())
    print("        x after", x[:, i].cpu().numpy(), "qw", float(ctrl[_capi.C_QW, i]), "t", float(ctrl[_capi.C_T, i]), "done", float(ctrl[_capi.C_DONE, i]), "steps", float(ctrl[_capi.C_STEPS, i]))
    print("        ctrl before", prev[1][:, i].cpu().numpy())
np.save("gpurun_out/vol_bad.npy", np.array([prev[0][:, bad[:16]].cpu().numpy(), x[:, bad[:16]].cpu().numpy()], dtype=object), allow_pickle=True) if len(bad) else None
