"""Summary of the PMC passes of the per-step workload at a large batch (PMC_ENVS=n python3 scripts/pmc_workload.py under
rocprofv3 --pmc ..., separate passes): HBM bytes per env-step (gfx950 corrections calibrated on the copies of the same run, as
scripts/pmc_summarise.py does) and the issue activity of k_step.   usage: python scripts/pmc_large_batch.py <dir> <n_envs>"""
import csv
import glob
import json
import os
import sys
from statistics import mean, median

d, n = sys.argv[1], int(sys.argv[2])
CAL = 64 << 20


def rows(sub):
    f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        out.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        out[r["Counter_Name"]][r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return out


def pick(c, frag):
    return [k for k in c if frag in k][0]


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_sbr2_amd import build as _b  # noqa: E402
res = {"envs_per_launch": n, "library_source_hash": open(_b.HASH).read().strip() if os.path.exists(_b.HASH) else None}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    c = rows(sub)[name]
    copies = [v for k in c if "direct_copy" in k or "copyBuffer" in k for v in c[k].values() if v > 0]
    big = [v for v in copies if v * 1024 > 0.2 * CAL]
    factor = median(big) * 1024 / CAL                     # counter units per byte really moved by a 64 MiB copy
    ks = c[pick(c, "k_step<")]
    res[name] = {"factor": factor, "bytes_per_env_step_mean": mean(ks.values()) * 1024 / factor / n,
                 "bytes_per_env_step_median": median(ks.values()) * 1024 / factor / n, "launches": len(ks)}
sq = rows("pmc_sq")
k = pick(sq["SQ_WAVES"], "k_step<")
tot = {name: sum(sq[name][k].values()) for name in sq if k in sq[name]}
res["kernel"] = k.split("(")[0]
res["sq"] = {"valu_insts_per_wave": tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"],
             "active_inst_any_over_wave_cycles": tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
             "wait_any_over_wave_cycles": tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
             "wait_inst_any_over_wave_cycles": tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
             "active_inst_valu_over_busy_cycles": tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_BUSY_CYCLES"]}
res["hbm_bytes_per_env_step"] = res["FETCH_SIZE"]["bytes_per_env_step_mean"] + res["WRITE_SIZE"]["bytes_per_env_step_mean"]
print(json.dumps(res, indent=1))
