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
    s0, s05, s0c = _j(tag, "bench_config2_scheme0"), _j(tag, "bench_config5_scheme0"), _j(tag, "bench_cycle_scheme0")
    pmc, sb = _j(tag, "pmc_traffic"), _j(tag, "serial_bound")
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")) as f:
        ref = json.load(f)
    kt, ro, cy = pmc["kernel_trace"], pmc["rollout"], pmc["cycle"]
    r2, rd = c2["roofline"], drv["roofline"]
    cpu, f2 = c2["cpu_baseline"], r2["fp64_valu"]
    gpu_tests = "?"
    log = os.path.join(ROOT, "profiles", "%s_pytest_gpu.log" % tag)
    if os.path.exists(log):
        m = re.search(r"(\d+) passed", open(log).read())
        gpu_tests = m.group(1) if m else "?"
    nb = {n: _j(tag, "bench_config2_n%d" % n) for n in (32768, 131072, 262144)}
    uni, walk = _j(tag, "bench_config2_uniform"), _j(tag, "bench_config2_walk")
    big, bigd = r2["larger_batches"]["262144"], rd["larger_batches"]["262144"]
    fd = rd["fp64_valu"]["b5_steps_per_interval"]
    rows = [
        ("configs[2], `python bench.py` (%d steps = four episodes, resets and terminal calls inside; cfg.scheme = 1)" % c2["steps"],
         "**%.3ge9 env-steps/s**, %.2f µs per step; the same command with `--scheme 0` (ten RK4 substeps per interval, rounds 1-4) on the "
         "same box: %.3ge9, %.2f µs" % (c2["value"] / 1e9, c2["ms_per_step"] * 1e3, s0["value"] / 1e9, s0["ms_per_step"] * 1e3),
         "`%s_bench_config2.json`, `%s_bench_config2_scheme0.json`" % (tag, tag)),
        ("configs[2], `--steps 20 --warmup 5` (the driver's command; since round 6 the timed calls are %d-%d of an episode: ten anoxic, the "
         "double-step call 51, nine aerobic - until round 5 calls 5-24, all anoxic)" % (drv["config"]["timed_calls"][0], drv["config"]["timed_calls"][1] - 1),
         "%.3ge9 env-steps/s, %.2f µs per step, %.2f µs per launch by events" % (drv["value"] / 1e9, drv["ms_per_step"] * 1e3, rd["avg_launch_us"]),
         "`%s_bench_config2_driver_style.json`" % tag),
        ("**262 144 envs on the one GPU, timed INSIDE the same command** (`roofline.larger_batches[\"262144\"]`, a second handle, the same bracket, "
         "window and K; `%s`)" % bigd["kernel"],
         "the driver's command: %.1f µs per step = **%.3f of the prescribed roofline by the wall clock** (%.3f by events), %.3ge9 env-steps/s; "
         "`python bench.py` (four whole episodes, resets and terminal calls inside): %.1f µs = %.3f (%.3f by events), %.3ge9"
         % (bigd["ms_per_step"] * 1e3, bigd["frac_wall"], bigd["frac_timed_launches"], bigd["env_steps_per_s"] / 1e9,
            big["ms_per_step"] * 1e3, big["frac_wall"], big["frac_timed_launches"], big["env_steps_per_s"] / 1e9),
         "`%s_bench_config2_driver_style.json`, `%s_bench_config2.json`" % (tag, tag)),
        ("`k_step<float,float,256,false,1,1>` per launch",
         "%.2f µs (`rocprofv3 --kernel-trace --stats`, %d calls, whole episodes), %.2f µs by events over the %d timed launches"
         % (kt["average_ns"] / 1e3, kt["calls"], r2["avg_launch_us"], r2["launches_timed"]), "`%s_bench_config2_kernel_stats.csv`" % tag),
        ("prescribed roofline (513 B × 65 536 per launch ÷ 8 TB/s)",
         "`roofline.frac` = **%.3f** (the conservative figure: `%s`); by the rocprof average of whole episodes %.3f, by the wall clock of "
         "the four episodes %.3f, by events over their launches %.3f; the driver's 20 launches: frac %.3f (wall %.3f, events %.3f)"
         % (r2["frac"], r2["frac_is"], r2["frac_episode"], r2["frac_wall"], r2["frac_timed_launches"], rd["frac"], rd["frac_wall"],
            rd["frac_timed_launches"]), "the three files above"),
        ("HBM traffic of `k_step` (PMC, gfx950-corrected)",
         "%.2f MB per launch = %.1f B per env-step (median %.1f) = %.3f × algorithmic; fetch %.2f MB, write %.2f MB"
         % (pmc["hbm_bytes_per_launch"] / 1e6, pmc["hbm_bytes_per_env_step"], pmc["hbm_bytes_per_env_step_median"],
            pmc["hbm_bytes_per_env_step"] / 513.0, pmc["fetch_corrected_bytes"] / 1e6, pmc["WRITE_SIZE_bytes"] / 1e6),
         "`%s_pmc_traffic.json`" % tag),
        ("`k_step` issue activity", "%.0f VALU instructions per wave and call (round 4, RK4: 3956); %s"
         % (pmc["valu_insts_per_wave"], pmc["sq_note"].split("; ", 1)[1]), "`%s_pmc_sq_by_kernel.csv`" % tag),
        ("what the integrator does on this workload, **counted on the device** (round 6: the plan row `SBR_C_PLAN` of the timed envs, read "
         "after each of those calls in one more, untimed episode of the same envs)",
         "%.2f Butcher-5 steps per interval and env, %.2f per wavefront (its slowest lane's count; the CPU oracle's sample of the workload: "
         "%.2f / %.2f); %.0f useful FLOP per env-step (RK4 × 10: %d); oxygen held (slaved) in %.0f %% of the env-calls; carbon dosed in %.0f %% "
         "of the wave-calls.  The driver's calls 41-60: %.2f / %.2f"
         % (f2["b5_steps_per_interval"]["per_env_mean"], f2["b5_steps_per_interval"]["per_wavefront_mean"],
            f2["b5_steps_per_interval"]["cpu_oracle_sample"]["per_env_mean"], f2["b5_steps_per_interval"]["cpu_oracle_sample"]["per_wavefront_mean"],
            f2["flop_per_env_step"], f2["b5_steps_per_interval"]["rk4_equivalent_flop"], 100 * f2["b5_steps_per_interval"]["slaved_share"],
            100 * c2["config"]["dosing_wave_call_share"], fd["per_env_mean"], fd["per_wavefront_mean"]),
         "`roofline.fp64_valu`, `config` of `%s_bench_config2.json`" % tag),
        ("the no-overlap bound (one wave per SIMD)",
         "memory %.2f µs at 8 TB/s + arithmetic %.2f µs (anoxic call %.2f, aerobic %.2f: per-wave medians of the stamp build) + empty "
         "dependent launch %.2f µs = %.2f µs = %.3f of the roofline"
         % (r2["serial_bound"]["memory_us"], r2["serial_bound"]["arithmetic_us_episode_mean"], sb["arithmetic_us"]["anoxic"],
            sb["arithmetic_us"]["aerobic"], sb["dependent_launch_floor_us"], r2["serial_bound"]["sum_us"],
            r2["serial_bound"]["frac_at_bound"]), "`%s_serial_bound.json`, `%s_step_timeline.log`" % (tag, tag)),
        ("fused rollout (configs[4])",
         "%.3ge10 env-steps/s (`--scheme 0`: %.3ge9); %.0f %% of the VALU issue slots; %.2f B per env-step moved (median of %d "
         "full-length launches; max %.2f)"
         % (c5["value"] / 1e10, s05["value"] / 1e9, 100 * c5["roofline"]["fp64_valu"]["issue_slot_frac"], ro["hbm_bytes_per_env_step"],
            ro["full_length_dispatches"], ro["hbm_bytes_per_env_step_max"]), "`%s_bench_config5.json`, `%s_pmc_traffic.json`" % (tag, tag)),
        ("per-cycle kernel (`SBR-v2`)",
         "%.3ge10 control intervals/s (`--scheme 0`: %.3ge10); %.0f %% of the issue slots; %.2f B per interval moved"
         % (cyc["value"] / 1e10, s0c["value"] / 1e10, 100 * cyc["roofline"]["fp64_valu"]["issue_slot_frac"], cy["hbm_bytes_per_env_step"]),
         "`%s_bench_cycle.json`" % tag),
        ("other batch sizes per launch (`--envs-per-gpu`; `roofline.larger_batches` of the default line)",
         "; ".join("%s envs: %.3ge9 env-steps/s, %.1f µs per step, %.3f of the prescribed roofline"
                   % ("{:,}".format(n).replace(",", " "), b["value"] / 1e9, b["ms_per_step"] * 1e3, b["roofline"]["frac"])
                   for n, b in nb.items()),
         ", ".join("`%s_bench_config2_n%d.json`" % (tag, n) for n in nb)),
        ("other policies, secondary lines (`--policy`; whole episodes, 65 536 envs)",
         "`uniform` (SURVEY §8d's synthetic inputs: U[0, 8] × U[0, 15] on all eight scenarios): %.3ge9 env-steps/s, %.1f µs per step, %.0f %% of "
         "the envs near a Monod pole at the end of an episode - under cfg.scheme = 1 the cost of a call depends on the states: %.2f × the physical "
         "policy's throughput; `walk` (the reference's own action model, `get_available_actions`): %.3ge9, %.2f µs per step, %.2f steps per "
         "wavefront (physical: %.2f)"
         % (uni["value"] / 1e9, uni["ms_per_step"] * 1e3, 100 * uni["env_status"]["near_pole_frac_last_episode"], uni["value"] / c2["value"],
            walk["value"] / 1e9, walk["ms_per_step"] * 1e3, walk["roofline"]["fp64_valu"]["b5_steps_per_interval"]["per_wavefront_mean"],
            f2["b5_steps_per_interval"]["per_wavefront_mean"]),
         "`%s_bench_config2_uniform.json`, `%s_bench_config2_walk.json`" % (tag, tag)),
        ("configs[1] (4 096 envs, fixed-step RK4 = cfg.scheme 0, 64 wavefronts: latency only)",
         "%.3ge8 env-steps/s, %.2f µs per launch" % (c1["value"] / 1e8, c1["roofline"]["avg_launch_us"]), "`%s_bench_config1.json`" % tag),
        ("CPU baseline on the GPU box (C port of the same algorithm, cfg.scheme = %d)" % cpu["scheme"],
         "%.3ge7 env-steps/s on %d threads, %.3ge6 on one" % (cpu["value"] / 1e7, cpu["cores"], cpu["single_thread"]["value"] / 1e6),
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
    s0 = _j(tag, "bench_config2_scheme0")
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")) as f:
        ref = json.load(f)
    kt, cpu = pmc["kernel_trace"], c2["cpu_baseline"]
    return ("* `SBROS-v1` — the step-level env (`SbrOS`): `sbr_reset` / `sbr_step` / fused `sbr_rollout`; **%.3ge9 env-steps/s** per step call at\n"
            "  65 536 envs on one MI355X (`python bench.py`, `profiles/%s_bench_config2.json`; with round 4's integrator, `--scheme 0`, on the same\n"
            "  box: %.3ge9), `k_step` %.2f µs per launch by the rocprof average of whole episodes = %.3f of the prescribed HBM roofline (round 4:\n"
            "  13.32 µs, 0.315); **262 144 envs on the same GPU, timed inside the same command: %.3f by the wall clock** (%.3ge9 env-steps/s); %.3ge10 in the\n"
            "  fused rollout (the Python reference: %.0f env-steps/s per core, `oracle/time_reference.py`; its C port %.2ge6 on one core of the GPU\n"
            "  box, %.2ge7 on %d)."
            % (c2["value"] / 1e9, tag, s0["value"] / 1e9, kt["average_ns"] / 1e3, c2["roofline"]["frac_episode"],
               c2["roofline"]["larger_batches"]["262144"]["frac_wall"], c2["roofline"]["larger_batches"]["262144"]["env_steps_per_s"] / 1e9,
               c5["value"] / 1e10, ref["value"], cpu["single_thread"]["value"] / 1e6, cpu["value"] / 1e7, cpu["cores"]))


def _replace_between(path, begin, end, text):
    s = open(path).read()
    if begin not in s or end not in s:
        raise SystemExit("%s has no %s ... %s markers" % (os.path.basename(path), begin, end))
    i, j = s.index(begin) + len(begin), s.index(end, s.index(begin))
    open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])


HL_BEGIN, HL_END = "<!-- headline:begin %s -->", "<!-- headline:end -->"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
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
