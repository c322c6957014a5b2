"""Register / scratch footprint of the kernels for a set of -D flags (cross-compiled, no GPU):
   python scripts/probes/regs.py "-DSBR_STEP_MIN_BLOCKS=2" [symbol-substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_sbr2_amd import build as B

flags = sys.argv[1].split() if len(sys.argv) > 1 else []
want = sys.argv[2:] or ["k_step"]
out = os.path.join(ROOT, "build", "regs_probe.s")
base = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
subprocess.check_call([B.hipcc()] + base + flags + ["-S", "--cuda-device-only", "-o", out, B.SRC], stderr=subprocess.DEVNULL)
asm = open(out).read()
for ent in re.split(r"\n  - \.agpr_count:", asm)[1:]:
    ent = ".agpr_count:" + ent
    name = re.search(r"\.name:\s+(\S+)", ent).group(1)
    if not any(w in name for w in want):
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, ent).group(1))
    m = re.search(r"^%s:.*?\n(.*?)\n\.Lfunc_end" % re.escape(name), asm, re.S | re.M)
    body = m.group(1) if m else ""
    ins = [l for l in body.split("\n") if l.strip() and not l.strip().endswith(":") and not l.strip().startswith((".", ";"))]
    print("%-40s vgpr %3d agpr %3d sgpr %3d scratch %4d B lds %6d  instr %5d  scratch-ops %d  accvgpr-moves %d" % (
        name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"),
        len(ins), sum("scratch_" in l for l in ins), sum("v_accvgpr" in l for l in ins)))
