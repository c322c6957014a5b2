"""Generates the "at a glance" table of DESIGN.md section 5 from the committed profile files, so that no figure in it can
differ from the JSON it quotes (the round-3 table said 1.36 B per env-step where the JSON held 3.51).

    python scripts/design_glance.py r04            prints the table
    python scripts/design_glance.py r04 --write    replaces the text between the glance markers of DESIGN.md

tests/test_bench_cpu.py regenerates the table and compares it with what DESIGN.md holds."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- glance:begin %s -->", "<!-- glance:end -->"


def _j(tag, name):
    with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))) as f:
        return json.load(f)


def table(tag):
    c2, drv, c5, cyc, c1 = (_j(tag, "bench_config2"), _j(tag, "bench_config2_driver_style"), _j(tag, "bench_config5"),
                            _j(tag, "bench_cycle"), _j(tag, "bench_config1"))
    pmc = _j(tag, "pmc_traffic")
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")) as f:
        ref = json.load(f)
    kt, ro, cy = pmc["kernel_trace"], pmc["rollout"], pmc["cycle"]
    r2, rd = c2["roofline"], drv["roofline"]
    cpu = c2["cpu_baseline"]
    gpu_tests = "?"
    log = os.path.join(ROOT, "profiles", "%s_pytest_gpu.log" % tag)
    if os.path.exists(log):
        m = re.search(r"(\d+) passed", open(log).read())
        gpu_tests = m.group(1) if m else "?"
    rows = [
        ("configs[2], `python bench.py` (%d steps = four episodes, resets and terminal calls inside)" % c2["steps"],
         "%.3ge9 env-steps/s, %.2f µs per step" % (c2["value"] / 1e9, c2["ms_per_step"] * 1e3), "`%s_bench_config2.json`" % tag),
        ("configs[2], `--steps 20 --warmup 5` (the driver's command)",
         "%.3ge9 env-steps/s, %.2f µs per step, %.2f µs per launch by events" % (drv["value"] / 1e9, drv["ms_per_step"] * 1e3, rd["avg_launch_us"]),
         "`%s_bench_config2_driver_style.json`" % tag),
        ("`k_step<float,float,256,false>` per launch",
         "%.2f µs (`rocprofv3 --kernel-trace --stats`, %d calls, whole episodes), %.2f µs by events over the %d timed launches"
         % (kt["average_ns"] / 1e3, kt["calls"], r2["avg_launch_us"], r2["launches_timed"]), "`%s_bench_config2_kernel_stats.csv`" % tag),
        ("prescribed roofline (513 B × 65 536 per launch ÷ 8 TB/s)",
         "**%.3f** by the rocprof average of whole episodes (`roofline.frac_episode`), %.3f by events over four episodes, %.3f over the "
         "driver's 20 launches (`roofline.frac`)" % (r2["frac_episode"], r2["frac"], rd["frac"]), "the three files above"),
        ("HBM traffic of `k_step` (PMC, gfx950-corrected)",
         "%.2f MB per launch = %.1f B per env-step (median %.1f) = %.3f × algorithmic; fetch %.2f MB, write %.2f MB"
         % (pmc["hbm_bytes_per_launch"] / 1e6, pmc["hbm_bytes_per_env_step"], pmc["hbm_bytes_per_env_step_median"],
            pmc["hbm_bytes_per_env_step"] / 513.0, pmc["fetch_corrected_bytes"] / 1e6, pmc["WRITE_SIZE_bytes"] / 1e6),
         "`%s_pmc_traffic.json`" % tag),
        ("`k_step` issue activity", "%.0f VALU instructions per wave and call; %s"
         % (pmc["valu_insts_per_wave"], pmc["sq_note"].split("; ", 1)[1]), "`%s_pmc_sq_by_kernel.csv`" % tag),
        ("fused rollout (configs[4])",
         "%.3ge9 env-steps/s; %.1f TFLOP/s = %.0f %% of the float64 vector peak (RK4 loops only), %.0f %% of the VALU issue slots; "
         "%.2f B per env-step moved (median of %d full-length launches; max %.2f)"
         % (c5["value"] / 1e9, c5["roofline"]["fp64_valu"]["achieved"], 100 * c5["roofline"]["fp64_valu"]["frac"],
            100 * c5["roofline"]["fp64_valu"]["issue_slot_frac"], ro["hbm_bytes_per_env_step"], ro["full_length_dispatches"],
            ro["hbm_bytes_per_env_step_max"]), "`%s_bench_config5.json`, `%s_pmc_traffic.json`" % (tag, tag)),
        ("per-cycle kernel (`SBR-v2`)",
         "%.3ge10 control intervals/s; %.0f %% of the float64 vector peak, %.0f %% of the issue slots; %.2f B per interval moved"
         % (cyc["value"] / 1e10, 100 * cyc["roofline"]["fp64_valu"]["frac"], 100 * cyc["roofline"]["fp64_valu"]["issue_slot_frac"],
            cy["hbm_bytes_per_env_step"]), "`%s_bench_cycle.json`" % tag),
        ("larger batches per launch (`--envs-per-gpu`; two and four waves per SIMD)",
         "; ".join("%s envs: %.3ge9 env-steps/s, %.1f µs per launch, %.3f of the prescribed roofline"
                   % ("{:,}".format(n).replace(",", " "), b["value"] / 1e9, b["roofline"]["avg_launch_us"], b["roofline"]["frac"])
                   for n, b in ((n, _j(tag, "bench_config2_n%d" % n)) for n in (131072, 262144))),
         "`%s_bench_config2_n131072.json`, `%s_bench_config2_n262144.json`" % (tag, tag)),
        ("configs[1] (4 096 envs, 64 wavefronts: latency only)",
         "%.3ge8 env-steps/s, %.2f µs per launch" % (c1["value"] / 1e8, c1["roofline"]["avg_launch_us"]), "`%s_bench_config1.json`" % tag),
        ("CPU baseline on the GPU box (C port of the same algorithm)",
         "%.3ge7 env-steps/s on %d threads, %.3ge5 on one" % (cpu["value"] / 1e7, cpu["cores"], cpu["single_thread"]["value"] / 1e5),
         "`cpu_baseline` of `%s_bench_config2.json`" % tag),
        ("the Python reference itself (build container, `oracle/time_reference.py`)",
         "%.0f env-steps/s on one core, %.0f on %d processes" % (ref["value"], ref["value_all_cores"], ref["cores_all"]),
         "`reference_cpu_timing.json`"),
        ("GPU tests", "%s passed" % gpu_tests, "`%s_pytest_gpu.log`" % tag),
    ]
    out = ["| quantity | value | where |", "|---|---|---|"]
    out += ["| %s | %s | %s |" % r for r in rows]
    return "\n".join(out)


def headline(tag):
    """README.md's first bullet, from the same files."""
    c2, c5, pmc = _j(tag, "bench_config2"), _j(tag, "bench_config5"), _j(tag, "pmc_traffic")
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")) as f:
        ref = json.load(f)
    kt, cpu = pmc["kernel_trace"], c2["cpu_baseline"]
    return ("* `SBROS-v1` — the step-level env (`SbrOS`): `sbr_reset` / `sbr_step` / fused `sbr_rollout`; **%.3ge9 env-steps/s** per step call at\n"
            "  65 536 envs on one MI355X (`python bench.py`, `profiles/%s_bench_config2.json`; round 3's driver record: 4.29e9), `k_step`\n"
            "  %.2f µs per launch by the rocprof average of whole episodes = %.3f of the prescribed HBM roofline (round 3: 14.84 µs,\n"
            "  0.283), and %.3ge9 in the fused rollout (the Python reference: %.0f env-steps/s per core, `oracle/time_reference.py`; its C\n"
            "  port %.2ge5 on one core of the GPU box, %.2ge7 on %d)."
            % (c2["value"] / 1e9, tag, kt["average_ns"] / 1e3, c2["roofline"]["frac_episode"], c5["value"] / 1e9, ref["value"],
               cpu["single_thread"]["value"] / 1e5, cpu["value"] / 1e7, cpu["cores"])).replace("\\n", "\n")


def _replace_between(path, begin, end, text):
    s = open(path).read()
    if begin not in s or end not in s:
        raise SystemExit("%s has no %s ... %s markers" % (os.path.basename(path), begin, end))
    i, j = s.index(begin) + len(begin), s.index(end, s.index(begin))
    open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])


HL_BEGIN, HL_END = "<!-- headline:begin %s -->", "<!-- headline:end -->"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    t = table(tag)
    if "--write" in sys.argv:
        _replace_between(os.path.join(ROOT, "DESIGN.md"), BEGIN % tag, END, t)
        _replace_between(os.path.join(ROOT, "README.md"), HL_BEGIN % tag, HL_END, headline(tag))
        print("DESIGN.md and README.md updated")
    else:
        print(t)
        print()
        print(headline(tag))


if __name__ == "__main__":
    main()
