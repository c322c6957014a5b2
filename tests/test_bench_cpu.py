"""Host logic of bench.py that needs no GPU: which committed PMC profile, if any, may be quoted in a run's JSON line."""
import json
import os
import warnings

from conftest import ROOT

import bench
from gym_sbr2_amd import build as B


def _write(d, name, h, **kw):
    rec = dict(library_source_hash=h, hbm_bytes_per_launch=38.8e6, hbm_bytes_per_env_step=592.5, envs_per_launch=65536, **kw)
    with open(os.path.join(d, name), "w") as f:
        json.dump(rec, f)


def test_pmc_profile_is_quoted_only_for_the_library_it_was_measured_on(tmp_path):
    d = str(tmp_path)
    h = B.source_hash()
    _write(d, "r07_pmc_traffic.json", h)
    rec, why = bench.pmc_record(h, d)
    assert why is None and rec["hbm_bytes_per_launch"] == 38.8e6 and rec["_file"].endswith("r07_pmc_traffic.json")
    # one flipped character of the recorded hash (the kernel changed, the profile was not refreshed): the figure disappears
    flipped = h[:-1] + ("0" if h[-1] != "0" else "1")
    _write(d, "r07_pmc_traffic.json", flipped)
    rec, why = bench.pmc_record(h, d)
    assert rec is None and h[:12] in why and "r07_pmc_traffic.json" in why
    # a library without a hash file (an A/B variant loaded through SBR_AMD_LIB) is never matched
    _write(d, "r07_pmc_traffic.json", h)
    assert bench.pmc_record(None, d)[0] is None
    # round 2's profile carries no hash at all: never matched either; the newest matching round wins
    _write(d, "r02_pmc_traffic.json", None)
    _write(d, "r05_pmc_traffic.json", h, marker=5)
    _write(d, "r06_pmc_traffic.json", flipped, marker=6)
    _write(d, "r07_pmc_traffic.json", flipped)
    assert bench.pmc_record(h, d)[0]["marker"] == 5
    assert bench.pmc_record(h, str(tmp_path / "nowhere"))[0] is None


def test_committed_profile_matches_the_sources_or_says_so():
    """Not an error (a kernel edit legitimately outdates the profile, and bench.py then reports traffic = null with the reason):
    a warning, so that the round's last profile refresh is not forgotten."""
    rec, why = bench.pmc_record(B.source_hash())
    if rec is None:
        warnings.warn("profiles/ holds no PMC profile of the current kernel sources: " + why)
    else:
        assert rec["hbm_bytes_per_launch"] > 0 and os.path.exists(os.path.join(ROOT, rec["_file"]))


def test_design_glance_table_is_generated_from_the_committed_profiles():
    """round 3's DESIGN quoted 1.36 B per env-step for the fused rollout where the committed JSON held 3.51.
    The "at a glance" table of DESIGN.md section 5 is now generated from the committed profile files
    (scripts/design_glance.py); regenerating it must give exactly the text DESIGN.md holds."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("design_glance", os.path.join(ROOT, "scripts", "design_glance.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    tag = "r06"
    begin = g.BEGIN % tag
    assert begin in text and g.END in text
    held = text[text.index(begin) + len(begin):text.index(g.END, text.index(begin))].strip()
    assert held == g.table(tag).strip()
    # README.md leads with figures of the same files: its first bullet is generated too
    readme = open(os.path.join(ROOT, "README.md")).read()
    hb = g.HL_BEGIN % tag
    assert hb in readme and g.HL_END in readme
    assert readme[readme.index(hb) + len(hb):readme.index(g.HL_END, readme.index(hb))].strip() == g.headline(tag).strip()
    # and the prose quotes the fused rollout's traffic as the JSON holds it (the figure round 3 got wrong)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)))
    assert ("%.2f B per env-step" % pmc["rollout"]["hbm_bytes_per_env_step"]) in text


def test_reference_cpu_timing_file_is_what_bench_reports():
    """cpu_baseline.reference is read from profiles/reference_cpu_timing.json, written by oracle/time_reference.py in the build
    container (the reference cannot travel): no constant pasted into bench.py."""
    rec = bench.reference_cpu()
    raw = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")))
    assert rec["value"] == raw["value"] and rec["cores"] == 1 and rec["value_all_cores"] == raw["value_all_cores"]
    assert 500 < rec["value"] < 1e4 and rec["script"] == "oracle/time_reference.py" and "NOT the GPU box" in rec["hardware"]
    assert abs(raw["one_process"]["episode_return_seed0"] - -0.8789670883455737) < 1e-12       # the reference's anchor, reproduced
    assert "REFERENCE_CPU" not in open(os.path.join(ROOT, "bench.py")).read()


def test_serial_bound_constants_and_larger_batches_are_hash_matched_committed_records(tmp_path):
    """VERDICT r4 items 2 and 6 / ADVICE r4: roofline.serial_bound's two measured constants and roofline.larger_batches come
    from committed files keyed by the library they were measured on; a kernel edit makes them disappear instead of going stale."""
    d = str(tmp_path)
    h = B.source_hash()
    flipped = h[:-1] + ("0" if h[-1] != "0" else "1")
    with open(os.path.join(d, "r05_serial_bound.json"), "w") as f:
        json.dump({"library_source_hash": h, "arithmetic_us": {"anoxic": 3.2, "aerobic": 4.9}, "dependent_launch_floor_us": 1.8}, f)
    rec = bench.serial_bound_record(h, d)
    assert rec["arithmetic_us"]["aerobic"] == 4.9 and rec["_file"].endswith("r05_serial_bound.json")
    assert bench.serial_bound_record(flipped, d) is None and bench.serial_bound_record(None, d) is None
    for n, hh in ((131072, h), (262144, flipped)):
        with open(os.path.join(d, "r05_bench_config2_n%d.json" % n), "w") as f:
            json.dump({"value": 6e9, "ms_per_step": 0.02, "roofline": {"frac": 0.4},
                       "config": {"envs_per_gpu": n, "library_source_hash": hh}}, f)
    lb = bench.larger_batches(h, d)
    assert list(lb) == ["131072"] and lb["131072"]["frac"] == 0.4 and lb["131072"]["us_per_launch"] == 20.0
    assert bench.larger_batches(flipped, d).keys() == {"262144"} and bench.larger_batches(None, d) is None
    # ... and the kernel-trace summary of the same run, when the round took one, travels with the entry
    with open(os.path.join(d, "r05_bench_config2_n131072_kernel_stats.csv"), "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                '"void k_step<float, float, 256, false, 1, 2>(double*)",1000,16810598,16810.598,93.0,15000,40000,100.0\n'
                '"void k_reset<float, false, 512>(SbrPar)",4,1800000,450000.0,6.0,440000,460000,10.0\n')
    kt = bench.larger_batches(h, d)["131072"]
    assert kt["kernel_trace_calls"] == 1000 and abs(kt["kernel_trace_us_per_launch"] - 16.810598) < 1e-9
    assert abs(kt["frac_kernel_trace"] - 131072 * 513 / 16.810598e-6 / 8e12) < 1e-9 and kt["frac"] == 0.4
    # no literal of the old kind is left in bench.py
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"arithmetic_us": 6.3' not in src and "1.82" not in src
    # the committed record of this round, if it is of the current sources, has the shape bench.py reads
    cur = bench.serial_bound_record(h)
    if cur is not None:
        assert set(cur["arithmetic_us"]) == {"anoxic", "aerobic"} and 0.5 < cur["dependent_launch_floor_us"] < 5


def test_design_per_scenario_table_is_generated_from_the_committed_json():
    """DESIGN.md 4.3.1 quotes worst gate and max lambda*dt PER SCENARIO; the table is generated by
    scripts/analysis/scenario_gates.py from profiles/r05_scenario_gates.json (which that script computes from the fixtures and the
    C oracle), and the figures themselves are inside the bars the parity tests assert."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scenario_gates", os.path.join(ROOT, "scripts", "analysis", "scenario_gates.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_scenario_gates.json")))
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    held = text[text.index(g.BEGIN) + len(g.BEGIN):text.index(g.END)].strip()
    assert held == g.markdown(rec).strip()
    assert set(rec["per_scenario"]) == {str(s) for s in range(8)} and len(rec["episodes"]) == 24
    for s, v in rec["per_scenario"].items():
        assert v["open"] <= 0.6 and v["open_s1"] <= 0.3 and v["closed_tight"] <= 0.6 and v["closed_tight_s1"] <= 0.5, (s, v)
        assert v["lam_dt"] < 2.785 / 2.5                      # RK4 x 10: at least 2.5 x inside its stability interval everywhere
    # the bench's own workload (physical policy on scenarios 4..7) stays inside the model's domain for the whole episode
    assert all(rec["episodes"]["scn%d_phys" % s]["valid_calls"] == 463 and rec["episodes"]["scn%d_phys" % s]["domain_exit_call"] == -1
               for s in (4, 5, 6, 7))


def test_episode_schedule_and_the_timed_window():
    """which calls of an episode are anoxic / aerobic comes from the reference's own phase tests on the running
    time (bench.episode_schedule), pinned here to the reference-captured interval log; the driver's 20-step region is placed across
    the first anoxic -> aerobic boundary with the episode's own mix (until round 5 it was calls 5..24, all anoxic, and the
    constant bench.py used for the anoxic calls, [(0, 46), (235, 405)], was wrong)."""
    import numpy as np
    from conftest import golden
    sched = bench.episode_schedule()
    e = golden("sbros_const_2_5")
    assert len(sched) == 463 == bench.CALLS_PER_EPISODE
    per_call = [tuple(int(k) for k in e["iv_kind"][e["iv_call"] == c]) for c in range(463)]
    assert sched == per_call                                             # every interval of every call, double steps included
    assert [c for c, k in enumerate(sched) if len(k) == 2] == [51, 275, 462] and sched[51] == (0, 1) and sched[275] == (1, 0)
    from gym_sbr2_amd import _capi
    assert bench.episode_schedule(_capi.default_config()) == sched       # ... also from the library's own default config
    anoxic = sum(1 for k in sched if k[-1] == 0)
    assert anoxic == 238
    # the driver's command: 20 timed calls after 5 warm-up calls -> calls 41 .. 60, ten of them anoxic
    assert bench.timed_window_start(20, 5, sched) == 41
    w = range(41, 61)
    assert sum(1 for c in w if sched[c][-1] == 0) == 10 and 51 in w
    # regions of an episode or more start at call W, as before; tiny regions still contain the boundary call
    assert bench.timed_window_start(1852, 50, sched) == 50 and bench.timed_window_start(463, 50, sched) == 50
    assert bench.timed_window_start(2, 0, sched) == 50 and bench.timed_window_start(400, 5, sched) == 5
    for k in (1, 2, 5, 20, 64, 100, 300, 462):
        s0 = bench.timed_window_start(k, 5, sched)
        assert 0 <= s0 <= 463 - k and (k < 2 or s0 <= 51 < s0 + k)      # (a region that nearly fills the episode starts the warm-up in the episode before)
    # no stale text: the docstring no longer calls config2 "RK4 h = dt", nor the uniform policy's cost data-independent
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "per-step API, RK4 h = dt" not in src and "cost is data-independent" not in src and "arithmetic cost is the same either way" not in src


def test_walk_policy_is_the_reference_action_model():
    """--policy walk: get_available_actions (gym_SBR_oneshot.py:440-459) - deltas (-0.1, 0, +0.1) / (-5, 0, +5) inside [0, 8] x [0, 15],
    from u_DO = 0, u_EC = 15 (:212-213); a move that is not available leaves the set-point where it is."""
    import numpy as np
    rs = np.random.RandomState(0)
    cur = np.column_stack([np.zeros(1000), np.full(1000, 15.0)])
    seen_do, seen_ec = set(), set()
    for _ in range(463):
        nxt = bench.walk_move(cur, rs.randint(0, 3, (1000, 2)), np)
        d = np.round(nxt - cur, 10)
        seen_do |= set(d[:, 0].tolist()); seen_ec |= set(d[:, 1].tolist())
        assert nxt[:, 0].min() >= 0 and nxt[:, 0].max() <= 8 and nxt[:, 1].min() >= 0 and nxt[:, 1].max() <= 15
        cur = nxt
    assert seen_do == {-0.1, 0.0, 0.1} and seen_ec == {-5.0, 0.0, 5.0}
    assert set(np.unique(cur[:, 1]).tolist()) <= {0.0, 5.0, 10.0, 15.0}


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch, capsys):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment spawns
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <args>` as a
    CHILD process (never an exec, before torch or a GPU is touched) and relays rank 0's JSON line and the return code."""
    import subprocess
    import sys
    import pytest
    calls = {}

    class Done:
        returncode = 0
        stdout = 'NCCL version banner\n{"metric": "env-steps/sec (batched)", "n_gpus": 2, "config": {"ranks": 2}}\n'

    def fake_run(cmd, **kw):
        calls["cmd"], calls["kw"] = cmd, kw
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "execv", lambda *a: (_ for _ in ()).throw(AssertionError("exec")))
    before = set(sys.modules)
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert ex.value.code == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    assert calls["kw"]["env"]["SBR_BENCH_SELF_LAUNCHED"] == "1" and "WORLD_SIZE" not in calls["kw"]["env"]
    out = capsys.readouterr()
    lines = [l for l in out.out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["config"]["ranks"] == 2 and "NCCL version banner" in out.err
    # a child that fails, or prints no line, is an error of this command too
    Done.returncode, Done.stdout = 3, ""
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "4"])
    assert ex.value.code == 3
    Done.returncode = 0
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "4"])
    assert ex.value.code == 1
    # under a launcher (WORLD_SIZE set) nothing is spawned: the rank count must match --gpus
    monkeypatch.setenv("WORLD_SIZE", "3")
    calls.clear()
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "2"])
    assert "cmd" not in calls and "WORLD_SIZE=3" in str(ex.value.code)
