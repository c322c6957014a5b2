"""Round 6, VERDICT r5 item 4, the device half of scripts/analysis/draw_sweep.py: whole episodes of the fused rollout (on-device
uniform policy U[0, 8] x U[0, 15], scenario = id mod 8, Philox influent noise) over tens of millions of envs, looking for envs
whose reactor volume after the done call is >= WV; every such env is replayed on the CPU oracle with the same Philox streams
(global env id, policy seed) and must come out the same - the device's draw is then the oracle's draw, garbage state or not.
Run on the GPU box:   python scripts/analysis/draw_sweep_gpu.py [--envs 50000000] > gpurun_out/draw_sweep_gpu.json"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import SbrOSVec, _capi  # noqa: E402
from gym_sbr2_amd.vec_env import load_influent_tables  # noqa: E402
from oracle import sbr_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=50_000_000)
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--max-found", type=int, default=40)
    args = ap.parse_args()
    means, stds = load_influent_tables()
    n = args.batch
    found, done, flagged, t0, k = [], 0, 0, time.time(), 0
    p = O.default_params()
    while done < args.envs and len(found) < args.max_found:
        first = k * n
        env = SbrOSVec(n, first_env_id=first, out_dtype=torch.float64)
        scen = ((torch.arange(n, device="cuda") + first) % 8).to(torch.int32)
        seed, pseed = 9000 + k, 77 + k
        env.reset(seed=seed, scenario=scen)
        env.rollout(463, policy_seed=pseed)
        x, ctrl = env.get_state()
        v = x[0]
        st = ctrl[_capi.C_STATUS].to(torch.int64)
        flagged += int(((st & 2) != 0).sum().item())
        bad = torch.nonzero(~(v < env.cfg.WV)).flatten().cpu().numpy()
        for i in bad[:8]:
            gid = first + int(i)
            one = O.OracleBatch(1, p, nthreads=1, first_env_id=gid)
            one.reset(one.mix(means, stds, np.array([gid % 8], dtype=np.int32), one.normals(seed)))
            one.rollout(463, pseed)
            xd = x[:, i].cpu().numpy()
            found.append({"global_env_id": gid, "reset_seed": seed, "policy_seed": pseed, "status_device": int(st[i].item()),
                          "status_oracle": int(one.envs["status"][0]), "V_device": float(xd[0]), "V_oracle": float(one.envs["x"][0][0]),
                          "qw_device": float(ctrl[_capi.C_QW][i].item()), "qw_oracle": float(one.envs["qw"][0]),
                          "x_device": xd.tolist(), "x_oracle": one.envs["x"][0].tolist()})
        env.close()
        done += n
        k += 1
        if k % 20 == 0:
            print("%d envs, %.0f s, %d found" % (done, time.time() - t0, len(found)), file=sys.stderr, flush=True)
    print(json.dumps({"envs": done, "near_pole_envs": flagged, "seconds": time.time() - t0, "instances": found}, indent=1))


if __name__ == "__main__":
    main()
