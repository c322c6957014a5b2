"""Workload for the rocprofv3 PMC passes: device-to-device calibration copies of known size, then the default bench
workload (config2, one episode) and the fused rollout (config5, one episode).  Run as `rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 scripts/pmc_workload.py`
(python3 directly after `--`; FETCH_SIZE and WRITE_SIZE in SEPARATE passes)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

CAL_BYTES = 64 << 20                     # the copy kernel reads and writes exactly this many bytes
src = torch.empty(CAL_BYTES, dtype=torch.uint8, device="cuda").fill_(1)
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(8):
    dst.copy_(src)
torch.cuda.synchronize()
print("calibration: 8 device-to-device copies of %d bytes" % CAL_BYTES, file=sys.stderr)
# per-step kernel, the fused rollout (one launch = 463 calls), the per-cycle kernel (one launch = 528 control intervals)
# PMC_ENVS=n: the per-step workload alone at n envs per launch (e.g. 262144: the two-waves-per-SIMD build of k_step)
envs = os.environ.get("PMC_ENVS")
for workload, steps in ((("config2", "463"),) if envs else (("config2", "463"), ("config5", "463"), ("cycle", "4"))):
    sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-large-leg", "--steps", steps, "--warmup", "20", "--workload", workload] + (
        ["--envs-per-gpu", envs] if envs else [])
    runpy.run_path(sys.argv[0], run_name="__main__")
