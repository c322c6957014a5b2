"""Per-phase means of the counters of a rocprofv3 --pmc pass over scripts/probes/pmc_phase.py (second episode only):
usage: python scripts/probes/pmc_phase_summary.py path/to/counter_collection.csv"""
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "k_step" in r["Kernel_Name"]:
            d = rows[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(rows)[-463:]
phases = {"anoxic 5..44": range(5, 45), "aerobic 60..220": range(60, 220), "anoxic 250..400": range(250, 400)}
names = sorted({k for i in ids for k in rows[i]})
print("%-18s" % "phase" + "".join("%22s" % n for n in names))
for ph, rg in phases.items():
    sel = [rows[ids[j]] for j in rg]
    print("%-18s" % ph + "".join("%22.1f" % (sum(s.get(n, 0.0) for s in sel) / len(sel)) for n in names))
