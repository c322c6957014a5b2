"""Output rows of sbr_step (float32 and float64 handles) against the state vector recomputed on the host from sbr_get_state:
state = [t, x] / x_1_state (gym_SBR_oneshot.py:153).  Prints which (row, column) entries are off, for three batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gym_sbr2_amd import SbrOSVec, _capi
X1 = np.array([0.5, 1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10])
for n in (64, 4096, 65536):
    for dt in (torch.float32, torch.float64):
        e = SbrOSVec(n, out_dtype=dt)
        scen = (torch.arange(n, device="cuda") % 8).to(torch.int32)
        e.reset(seed=1, scenario=scen)
        a = torch.rand(n, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
        for c in range(2):
            o, s, r, d = e.step(a)
            x, ctrl = e.get_state()
            torch.cuda.synchronize()
            want = np.concatenate([ctrl[_capi.C_T].cpu().numpy()[:, None], x.cpu().numpy().T], axis=1) / X1
            got = s.double().cpu().numpy()
            bad = np.argwhere(np.abs(got - want) > 1e-6 * (1 + np.abs(want)))
            print(n, str(dt)[6:], "call", c, "state entries off:", len(bad), bad[:10].tolist(), flush=True)
        e.close()
