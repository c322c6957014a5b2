"""Collects the two measured constants of bench.py's roofline.serial_bound ON THE GPU BOX and writes them, keyed by the hash of
the library sources they were measured on, to profiles/<tag>_serial_bound.json (ADVICE r4 / VERDICT r4 item 2: they used to be
literals in bench.py).

  arithmetic_us              median over waves of the stamp segment "before the intervals -> PIDs + integration done" of one
                             k_step launch of 65 536 envs in a back-to-back sequence (diagnostic build -DSBR_STAMPS,
                             scripts/probes/step_timeline.py), for an anoxic and an aerobic call
  dependent_launch_floor_us  launch-to-launch period of the same kernel returning at its first instruction, replayed from a
                             captured graph (scripts/probes/boundary_floor.py): what a dependent launch costs without any work

usage (inside a gpurun call): python scripts/serial_bound.py r05"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import build as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
stamps = os.path.join(ROOT, "build", "libsbr_amd_stamps.so")
os.makedirs(os.path.dirname(stamps), exist_ok=True)
flags = [f for f in B.FLAGS] + ["-DSBR_STAMPS"]
subprocess.check_call([B.hipcc()] + flags + ["-o", stamps, B.SRC])          # always rebuilt: the same sources as the library
env = dict(os.environ, SBR_AMD_LIB=stamps, SBR_AMD_ALLOW_ABI_MISMATCH="0")
tl = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probes", "step_timeline.py"), "65536"], env=env,
                    capture_output=True, text=True).stdout
fl = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probes", "boundary_floor.py")], env=env,
                    capture_output=True, text=True).stdout
seg = re.findall(r"N = 65536, (anoxic|aerobic)[^\n]*\n(?:.*\n)*?\s+per-wave segment medians: .*?2->3 ([0-9.]+)", tl)
life = re.findall(r"\s+7 stores acknowledged\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)", tl)
floor = re.search(r"N =\s+65536:.*the same kernel returning at once: [0-9.]+ us eager \(host-bound: Python \+ launch\), ([0-9.]+) us in a graph", fl)
assert len(seg) == 2 and floor, (tl[-2000:], fl[-800:])
rec = {"library_source_hash": B.source_hash(),
       "envs_per_launch": 65536,
       "arithmetic_us": {k: float(v) for k, v in seg},
       "wave_lifetime_us_median": {k: float(l[1]) for (k, _), l in zip(seg, life)},
       "dependent_launch_floor_us": float(floor.group(1)),
       "policy": "step_timeline.py: u_DO ~ U[0, 8], u_EC ~ U[0, 15] drawn once (every wavefront holds a lane in the oxygen knee in "
                 "the aerobic phase: four Butcher-5 steps; the anoxic call: two)",
       "source": "scripts/serial_bound.py: scripts/probes/step_timeline.py (stamp build of the same sources) and "
                 "scripts/probes/boundary_floor.py on MI355X",
       "logs": {"step_timeline": "profiles/%s_step_timeline.log" % tag, "boundary_floor": "profiles/%s_boundary_floor.log" % tag}}
with open(os.path.join(ROOT, "profiles", "%s_step_timeline.log" % tag), "w") as f:
    f.write(tl)
with open(os.path.join(ROOT, "profiles", "%s_boundary_floor.log" % tag), "w") as f:
    f.write(fl)
with open(os.path.join(ROOT, "profiles", "%s_serial_bound.json" % tag), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
