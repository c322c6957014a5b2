"""Distils gpurun_out/<tag>/ (written by scripts/profile_round.sh, which also runs this on the GPU box) into profiles/:
    <tag>_bench_<workload>.json, <tag>_bench_config2_kernel_stats.csv, <tag>_pmc_{fetch,write,sq}_by_kernel.csv,
    <tag>_pmc_traffic.json  (HBM bytes per k_step launch, read by bench.py as roofline.traffic)
Unit handling as /opt/skills/guides/MI355X_MICROARCH.md prescribes: counter values are KiB; WRITE_SIZE is exact; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is corrected with the factor measured IN THE SAME PASS on
device-to-device copies of known size (scripts/pmc_workload.py).   usage: python scripts/pmc_summarise.py r01"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
CAL_BYTES = 64 << 20


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


DURATIONS = {}      # pass_dir -> {kernel: [duration in us, in the order of the counter values]}


def counters(pass_dir):
    """{counter: {kernel: [values]}} of one PMC pass, in dispatch order; the durations of the same dispatches (from the
    Start/End timestamps rocprofv3 writes next to every counter value) go to DURATIONS[pass_dir][kernel]"""
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(dict)
    with open(one(pass_dir + "/**/*counter_collection.csv")) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        vals[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            durs[r["Kernel_Name"]][int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    DURATIONS[pass_dir] = {k: [d[i] for i in sorted(d)] for k, d in durs.items()}
    return vals


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def full_length(pass_dir, kernel, values):
    """Indices of the FULL-LENGTH launches of a fused kernel (the warm-up launches of bench.py run 19 calls or 1, the timed and
    priming ones 463): dispatches that last at least 80 % of the median of the longer half.  Round 3 took max() over all
    dispatches instead, which picked one outlier (see `preempted` below)."""
    d = DURATIONS.get(pass_dir, {}).get(kernel)
    if not d or len(d) != len(values):
        ref = median(sorted(values)[len(values) // 2:])          # no timestamps: fall back to the counter itself
        return [i for i, v in enumerate(values) if v >= 0.8 * ref]
    ref = median(sorted(d)[len(d) // 2:])
    return [i for i, x in enumerate(d) if x >= 0.8 * ref]


def robust(pass_dir, kernel, values):
    """median / max / count of a counter over the full-length launches + the dispatches that stand out (> 1.5 x median)"""
    idx = full_length(pass_dir, kernel, values)
    sel = [values[i] for i in idx]
    med = median(sel)
    d = DURATIONS.get(pass_dir, {}).get(kernel)
    out = [{"dispatch_index": i, "value": values[i], "duration_us": d[i] if d and len(d) == len(values) else None,
            "median_duration_us": median([d[j] for j in idx]) if d and len(d) == len(values) else None}
           for i in idx if values[i] > 1.5 * med]
    return med, max(sel), len(sel), out


def short(k):
    return k.split("(")[0][:60]


def by_kernel_csv(path, name, table):
    with open(path, "w") as f:
        f.write("kernel,dispatches,mean_%s,min,max\n" % name)
        for k, v in sorted(table.items(), key=lambda kv: -sum(kv[1]))[:12]:
            f.write('"%s",%d,%.1f,%.1f,%.1f\n' % (short(k), len(v), sum(v) / len(v), min(v), max(v)))


for w in ("config2", "config1", "config5", "cycle"):
    p = os.path.join(src, "bench_%s.json" % w)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_bench_%s.json" % (tag, w)))
shutil.copy(one("trace_config2/**/*kernel_stats.csv"), os.path.join(dst, "%s_bench_config2_kernel_stats.csv" % tag))
KERNEL_TRACE = None          # the k_step row of the --kernel-trace --stats run (whole episodes): bench.py's roofline.frac_episode
with open(os.path.join(dst, "%s_bench_config2_kernel_stats.csv" % tag)) as f:
    for r in csv.DictReader(f):
        if "k_step<" in r["Name"]:
            KERNEL_TRACE = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                            "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "percentage_of_gpu_time": float(r["Percentage"]),
                            "file": "profiles/%s_bench_config2_kernel_stats.csv" % tag,
                            "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-large-leg (1852 steps = four "
                                       "episodes incl. their terminal calls; priming and warm-up launches included)"}
            break

fetch, write = counters("pmc_fetch")["FETCH_SIZE"], counters("pmc_write")["WRITE_SIZE"]
by_kernel_csv(os.path.join(dst, "%s_pmc_fetch_by_kernel.csv" % tag), "FETCH_SIZE_KiB", fetch)
by_kernel_csv(os.path.join(dst, "%s_pmc_write_by_kernel.csv" % tag), "WRITE_SIZE_KiB", write)


def pick(table, needle):
    ks = [k for k in table if needle in k]
    if not ks:
        raise SystemExit("no kernel matching %r" % needle)
    return max(ks, key=lambda k: len(table[k]))


def mean(v):
    return sum(v) / len(v)


step_k = pick(fetch, "k_step<")
cal_f = [v for k in fetch if "copyBuffer" in k for v in fetch[k] if v * 1024 > 0.25 * CAL_BYTES]
cal_w = [v for k in write if "copyBuffer" in k for v in write[k] if v * 1024 > 0.5 * CAL_BYTES]
if len(cal_f) < 4 or len(cal_w) < 4:
    raise SystemExit("calibration copies not found in the PMC passes (%d, %d)" % (len(cal_f), len(cal_w)))
f_factor = mean(cal_f) * 1024 / CAL_BYTES          # ~0.50 on gfx950
w_factor = mean(cal_w) * 1024 / CAL_BYTES          # ~1.00
n_envs = 65536
sys.path.insert(0, ROOT)
from gym_sbr2_amd import build as _build  # noqa: E402
hash_file = os.path.join(src, "library_source_hash.txt")
lib_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else None
if lib_hash != _build.source_hash():
    print("WARNING: the profiled library (%s) is not what the source tree builds now (%s): bench.py will not attach this profile"
          % (str(lib_hash)[:12], _build.source_hash()[:12]))
fetch_raw = mean(fetch[step_k]) * 1024
fetch_b = fetch_raw / f_factor
write_b = mean(write[pick(write, "k_step<")]) * 1024 / w_factor
out = {
    "source": "scripts/profile_round.sh %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (each with "
              "--kernel-trace only) over scripts/pmc_workload.py = calibration copies + `bench.py --no-cpu-baseline --steps 463` "
              "on MI355X; MEANS over the k_step dispatches" % tag,
    "unit_note": "counter values are KiB.  Both counters are calibrated in their own pass on eight %d-byte device-to-device "
                 "copies (__amd_rocclr_copyBuffer): WRITE_SIZE/bytes = %.4f, FETCH_SIZE/bytes = %.4f (gfx950 tallies 128-byte "
                 "fetches at 64 bytes, MI355X_MICROARCH.md HBM section)" % (CAL_BYTES, w_factor, f_factor),
    "library_source_hash": lib_hash, "policy": "physical",
    "envs_per_launch": n_envs, "kernel": short(step_k), "dispatches": len(fetch[step_k]),
    "FETCH_SIZE_raw_bytes": fetch_raw, "fetch_calibration_factor": f_factor, "write_calibration_factor": w_factor,
    "fetch_corrected_bytes": fetch_b, "WRITE_SIZE_bytes": write_b,
    "hbm_bytes_per_launch": fetch_b + write_b, "hbm_bytes_per_env_step": (fetch_b + write_b) / n_envs,
    "hbm_bytes_per_env_step_median": (median(fetch[step_k]) * 1024 / f_factor + median(write[pick(write, "k_step<")]) * 1024 / w_factor) / n_envs,
    "hbm_bytes_per_env_step_max": (max(fetch[step_k]) * 1024 / f_factor + max(write[pick(write, "k_step<")]) * 1024 / w_factor) / n_envs,
    "expected_from_code": {"read_bytes_per_env": 264,
                           "write_bytes_per_env": "313 when no lane of the wave dosed carbon (V, Si, Xi not stored), 337 otherwise",
                           "note": "x 112 R; 18 ctrl rows R = 144 (t, So[-1], Sno[-1], 2 integrals, EC[-1], return, meta, 10 ring "
                                   "slots); action 8 R; x 88-112 W; 11 ctrl rows W = 88; obs 72 + state 60 + reward 4 + done 1 W"},
    "algorithmic_bytes_per_env_step": 513,
    "kernel_trace": KERNEL_TRACE,
}
try:                                     # the fused rollout: one launch = a whole episode of 463 calls kept in registers
    rk, rkw = pick(fetch, "k_rollout<"), pick(write, "k_rollout<")
    f_med, f_max, f_n, f_out = robust("pmc_fetch", rk, fetch[rk])
    w_med, w_max, w_n, w_out = robust("pmc_write", rkw, write[rkw])
    rf, rw = f_med * 1024 / f_factor, w_med * 1024 / w_factor            # MEDIAN over the full-length (463-call) launches
    out["rollout"] = {"kernel": short(rk), "dispatches": len(fetch[rk]), "full_length_dispatches": f_n, "calls_per_launch": 463,
                      "statistic": "median over the full-length launches (selected by duration); round 3 published max(), which "
                                   "was one preempted dispatch",
                      "fetch_bytes_median": rf, "fetch_bytes_max": f_max * 1024 / f_factor,
                      "write_bytes_median": rw, "write_bytes_max": w_max * 1024 / w_factor,
                      "hbm_bytes_per_launch": rf + rw, "hbm_bytes_per_env_step": (rf + rw) / n_envs / 463,
                      "hbm_bytes_per_env_step_max": (f_max * 1024 / f_factor + w_max * 1024 / w_factor) / n_envs / 463,
                      "outlier_dispatches": {"fetch": f_out, "write": w_out,
                                             "reading": "a full-length launch holds 1024 waves x 64 lanes x 256 VGPRs x 4 B = 67 MB of register state; "
                                                        "a dispatch that fetches ~67 MB (corrected) MORE than its peers and lasts ~1 ms longer was "
                                                        "preempted once and restored (compute wave save/restore): traffic of the context "
                                                        "switch, not of the kernel"},
                      "note": "plant and controller state are loaded once and stored once per launch; the per-step convention "
                              "would charge 463 x 513 B per env"}
    print("k_rollout: median %.2f MB per launch = %.2f B per env-step over %d full-length launches (max %.2f B; %d outlier dispatches)"
          % ((rf + rw) / 1e6, out["rollout"]["hbm_bytes_per_env_step"], f_n, out["rollout"]["hbm_bytes_per_env_step_max"],
             len(f_out) + len(w_out)))
except SystemExit:
    print("no k_rollout dispatches in the PMC passes")
try:                                     # the per-cycle kernel: one launch = 528 control intervals (+ its reset)
    ck, crk = pick(fetch, "k_cycle<"), pick(fetch, "k_cycle_reset<")
    cf = (mean(fetch[ck]) + mean(fetch[crk])) * 1024 / f_factor
    cw = (mean(write[pick(write, "k_cycle<")]) + mean(write[pick(write, "k_cycle_reset<")])) * 1024 / w_factor
    out["cycle"] = {"kernel": short(ck), "dispatches": len(fetch[ck]), "intervals_per_launch": 528,
                    "hbm_bytes_per_launch": cf + cw, "hbm_bytes_per_env_step": (cf + cw) / n_envs / 528}
    print("k_cycle + reset: %.2f MB per cycle = %.3f B per control interval" % ((cf + cw) / 1e6, out["cycle"]["hbm_bytes_per_env_step"]))
except SystemExit:
    print("no k_cycle dispatches in the PMC passes")
try:                                     # dynamic VALU instructions per wave of k_step (SQ pass), for bench.py's issue_slot_frac
    sq0 = counters("pmc_sq")
    sk = pick(sq0["SQ_INSTS_VALU"], "k_step<")
    out["valu_insts_per_wave"] = mean(sq0["SQ_INSTS_VALU"][sk]) / mean(sq0["SQ_WAVES"][sk])
    out["sq_note"] = ("SQ pass of the same workload: SQ_INSTS_VALU / SQ_WAVES of %s; SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.3f, "
                      "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f, SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f"
                      % (short(sk), mean(sq0["SQ_ACTIVE_INST_ANY"][sk]) / mean(sq0["SQ_WAVE_CYCLES"][sk]),
                         mean(sq0["SQ_WAIT_INST_ANY"][sk]) / mean(sq0["SQ_WAVE_CYCLES"][sk]),
                         mean(sq0["SQ_WAIT_ANY"][sk]) / mean(sq0["SQ_WAVE_CYCLES"][sk])))
    print("k_step: %.0f VALU instructions per wave" % out["valu_insts_per_wave"])
    for key, needle, extra in (("rollout", "k_rollout<", None), ("cycle", "k_cycle<", "k_cycle_reset<")):
        if key not in out:
            continue
        kk = pick(sq0["SQ_INSTS_VALU"], needle)
        fl = full_length("pmc_sq", kk, sq0["SQ_INSTS_VALU"][kk])
        per_wave = median([sq0["SQ_INSTS_VALU"][kk][i] for i in fl]) / mean(sq0["SQ_WAVES"][kk])       # the full-length launches
        if extra:
            ek = pick(sq0["SQ_INSTS_VALU"], extra)
            per_wave += mean(sq0["SQ_INSTS_VALU"][ek]) / mean(sq0["SQ_WAVES"][ek])
        out[key]["valu_insts_per_wave"] = per_wave
        i_full = fl[len(fl) // 2]
        out[key]["sq_active_inst_any_over_wave_cycles"] = sq0["SQ_ACTIVE_INST_ANY"][kk][i_full] / sq0["SQ_WAVE_CYCLES"][kk][i_full]
        out[key]["sq_wait_any_over_wave_cycles"] = sq0["SQ_WAIT_ANY"][kk][i_full] / sq0["SQ_WAVE_CYCLES"][kk][i_full]
        print("%s: %.0f VALU instructions per wave and launch" % (needle, per_wave))
except (SystemExit, KeyError, ZeroDivisionError):
    print("no SQ pass: valu_insts_per_wave not recorded")
json.dump(out, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print("k_step: fetch %.2f MB + write %.2f MB = %.1f B per env-step (factors %.4f / %.4f)"
      % (fetch_b / 1e6, write_b / 1e6, out["hbm_bytes_per_env_step"], f_factor, w_factor))

try:
    sq = counters("pmc_sq")
    with open(os.path.join(dst, "%s_pmc_sq_by_kernel.csv" % tag), "w") as f:
        names = sorted(sq)
        f.write("kernel,dispatches," + ",".join("mean_" + n for n in names) + "\n")
        kernels = sorted({k for n in names for k in sq[n]}, key=lambda k: -len(sq[names[0]].get(k, [])))[:8]
        for k in kernels:
            f.write('"%s",%d,' % (short(k), len(sq[names[0]].get(k, []))) +
                    ",".join("%.1f" % mean(sq[n][k]) if sq[n].get(k) else "" for n in names) + "\n")
except SystemExit:
    print("no SQ pass")
