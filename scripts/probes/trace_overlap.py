"""Do the k_step launches of two streams overlap in time?  Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and
prints, for the k_step dispatches: count, mean duration, the union of their busy intervals, the time during which two or more
were in flight, and the mean start-to-start period per queue.
usage: python scripts/probes/trace_overlap.py path/to/kernel_trace.csv [skip_first_n]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "k_step" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2        # the second half: steady clocks, no resets in between
rows = rows[skip:]
if not rows:
    raise SystemExit("no k_step dispatches in the trace")
dur = [b - a for a, b, _ in rows]
events = sorted([(a, 1) for a, _, _ in rows] + [(b, -1) for _, b, _ in rows])
depth, last, busy, multi = 0, events[0][0], 0, 0
for t, d in events:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
span = rows[-1][1] - rows[0][0]
queues = sorted({q for _, _, q in rows})
print("k_step dispatches analysed: %d on queues %s" % (len(rows), queues))
print("mean duration %.2f us (min %.2f, max %.2f)" % (sum(dur) / len(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3))
print("span %.1f us: some k_step in flight %.1f %% of it, two or more in flight %.1f %%" % (span / 1e3, 100.0 * busy / span, 100.0 * multi / span))
for q in queues:
    st = [a for a, _, qq in rows if qq == q]
    if len(st) > 1:
        print("queue %s: %d launches, mean start-to-start period %.2f us" % (q, len(st), (st[-1] - st[0]) / (len(st) - 1) / 1e3))
print("period per step of all groups: %.2f us" % (span / 1e3 / (len(rows) / max(len(queues), 1))))
