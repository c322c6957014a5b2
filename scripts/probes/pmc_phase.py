"""One episode of bench.py's workload at 65536 envs for a PMC pass (rocprofv3 --pmc ... --kernel-trace -- python3 this.py):
the library comes from SBR_AMD_LIB, so two kernel variants can be compared counter by counter, phase by phase
(scripts/probes/pmc_phase_summary.py groups the k_step dispatches by their index in the episode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_sbr2_amd import SbrOSVec
N = 65536
env = SbrOSVec(N)
gid = torch.arange(N, device="cuda")
scen = (4 + gid % 4).to(torch.int32)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
pool = torch.rand(64, N, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
if os.environ.get("AB_POLICY") == "dose":
    pool[:, :, 1] = 0.0
for ep in range(2):
    env.reset(seed=2 + ep, scenario=scen)
    for j in range(463):
        env.step(pool[j & 63])
torch.cuda.synchronize()
env.close()
