#!/usr/bin/env python3
"""bench.py - batched SBROS-v1 env-steps/s on N MI355X (one process per GPU), next to the kernel's roofline and a
CPU baseline.

    python bench.py --gpus 1 --steps K --warmup W                         (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W                            (N > 1, RCCL)

A "step" is one sbr_step() launch over this rank's batch: every env advances one control interval (two on the three
phase-boundary calls of an episode).  Workloads (BASELINE.json `configs`):
    config2 (default)  65536 envs per GPU, stochastic influent (Philox normals drawn on the device), per-call random float32
                       set-points already resident in HBM, per-step API, cfg.scheme = 1 (adaptive Butcher-5 per interval and
                       env; `--scheme 0`: ten RK4 substeps, h = dt).  With N > 1 GPUs the envs are sharded by
                       global id through gym_sbr2_amd.ShardedSbrOS (the class the sharding tests cover); the one collective of
                       the path, an RCCL all-gather of the float32 episode returns, runs at every episode boundary (every 463
                       calls).  A timed region shorter than an episode (the driver's `--steps 20 --warmup 5`) contains no
                       boundary, so ONE all-gather is then issued after the K-th step, before the closing synchronise: the timed
                       region of an N > 1 run always contains the collective, and `config.allgathers_in_timed_region` /
                       `config.allgather_bytes_per_rank` say how many and how large.  The N = 1 workload is unchanged by this
                       (no process group, no collective).  configs[3]'s shape at 65536 envs per GPU (per-GPU work fixed =>
                       weak scaling; 8 GPUs = 524288 envs = 2 x configs[3]'s 262144; `--envs-per-gpu 32768` gives configs[3]
                       itself).
    config1            4096 envs per GPU, deterministic influent (64 wavefronts: cannot fill 1024 SIMDs; a parity case)
    config5            65536 envs per GPU, fused on-device random-policy rollout (sbr_rollout), 463 calls per launch
    cycle              65536 envs per GPU of the per-cycle env SBR-v2: one launch = one whole cycle of 528
                       control intervals; a "step" is then one cycle and `value` is still control intervals per second
Policies (`--policy`):
    physical (default) set-points u_DO ~ U[0, 2.5], u_EC ~ U[0, 15] per call on the four high-ammonia influent scenarios
                       (4 + global id mod 4; the reference's own SbrOS uses scenario 6): every env stays inside the model's
                       physical domain for the whole episode (`env_status.near_pole_frac_last_episode` = 0), so the timed
                       trajectories are ones on which parity with the reference is defined and asserted
                       (tests/test_gpu_parity.py::test_bench_workload_parity_at_65536_with_the_physical_policy).
    uniform            u_DO ~ U[0, 8], u_EC ~ U[0, 15] on all eight scenarios (SURVEY.md 8d's synthetic inputs, round 1's workload):
                       over-aerates, drives ammonia negative in 86 % of the envs (the reference model has no guards).  Under
                       cfg.scheme = 1 the cost of a call depends on the state (step counts are chosen per env): the measured
                       ratio to the physical policy is in profiles/r06_bench_config2_uniform.json / DESIGN.md section 5.
    walk               the policy shape the reference itself defines (get_available_actions, gym_SBR_oneshot.py:440-459): from
                       u_DO = 0, u_EC = 15 (:212-213) every call moves each set-point by one of {-0.1, 0, +0.1} / {-5, 0, +5},
                       uniformly among the moves that stay inside [0, 8] x [0, 15]; scenarios 4..7.  A SECONDARY line (steadier
                       set-points keep lanes out of the oxygen knee, so scheme 1 takes fewer steps): never the headline.
Which calls are timed.  A region shorter than an episode (the driver's `--steps 20 --warmup 5`) is placed so that it straddles the
first anoxic -> aerobic phase boundary with the episode's own mix of anoxic and aerobic calls (51 % anoxic): the episode is advanced
untimed to call 36, five warm-up calls follow, and calls 41 .. 60 are timed - ten anoxic calls (two-step intervals), the
phase-boundary call with its double step, nine aerobic ones (`config.timed_calls`, `config.anoxic_share_of_timed_calls`).  Until
round 5 the region was calls 5 .. 24, all anoxic: the cheapest twenty calls of an episode.  Regions of an episode or more start at
call W as before (they hold every kind of call, the terminal call and the reset).
The large-batch leg.  With `--gpus 1`, workload config2 and no `--envs-per-gpu`, the run then times a SECOND handle of 262 144 envs
(configs[3]'s total on one GPU, 84 MB of state; above 65 536 envs a launch has more wavefronts than the chip has SIMDs and runs the
two-waves-per-SIMD build of k_step) with the same bracket, priming, window and K, and reports it as
`roofline.larger_batches["262144"]` with `measured_in_this_run: true`: north_star's ">= 40 % of HBM roofline on one MI355X" measured
inside this very command.  The headline fields stay configs[2]'s.  `--no-large-leg` skips it.
`python bench.py --gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset) starts the ranks itself: a child process
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py ...`
(subprocess, started before anything in this process touches a GPU; never an exec), whose one JSON line and return code are relayed.
Episodes end after 463 calls; the reset (influent draw + 252-substep fill phase) runs INSIDE the timed region and is
not counted as steps.  Before the W warm-up steps the same workload runs untimed for PRIME_SECONDS of wall time: the GPU needs
~25 ms of sustained work to reach its steady clocks (measured with scripts/probes/clock_ramp.py: 20.95 us per launch in the
first block, 19.5 us from the fourth on, 21.1 us again after 2 s idle), which a warm-up of a few dozen 20-us steps never gives.  `value` = (envs of all ranks) * K / (max over ranks of the wall time of the K steps).
The K steps are bracketed by barrier + torch.cuda.synchronize() on both sides; each rank reads its clock right after its own
closing synchronise (before the closing barrier), and the MAX over ranks is taken: the time until the slowest rank has finished
its K steps, without the latency of the closing collective itself.  The opening bracket is synchronise, barrier, the LAST of the
W warm-up steps, synchronise: the first event record and the first launch that follow an RCCL collective cost the host 30-45 us
each instead of 4-5 (measured with one rank, profiles/r02_notes.md), a cost of the harness's barrier that a 20-step region
(285 us of kernels) would otherwise carry as 70-80 us; ranks therefore start within one step (14 us) of each other.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # the platform default; host-resident kernargs cost +4 us per launch

ALGO_BYTES_PER_ENV_STEP = 513          # x 112+112, ctrl 72+72, action 8, obs 72, state 60, reward 4, done 1
HBM_PEAK_GBPS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
CALLS_PER_EPISODE = 463
# float64 operations per RK4 substep, counted in the gfx950 ISA of the two substep loops of k_step: when no lane of the wave
# doses carbon 155 FMA x 2 + 104 MUL + 8 ADD + 4 RCP = 426; with dosing (round 4: the scaled-mass loop, unrolled by two:
# 338 x 2 + 216 + 28 + 8 per two substeps) 464 (rounds 2-3, concentration form: 552; until the library was built with
# -ffp-contract=off the backend had turned four of the adds per substep into FMAs: 430 / 468 by this count, same instructions).  The loops only (no PIDs, reward,
# observations): a LOWER bound of the work per env-step.  tests/test_isa_cpu.py asserts both figures against the compiler's output.
FP64_FLOP_PER_SUBSTEP = {"plain": 426, "dosing": 464, "filling": 586}      # filling: k_cycle's / k_reset's loop, (418 x 2 + 229 + 96 + 12) / 2
SUBSTEPS = 10
# cfg.scheme = 1 (round 5, the default): float64 operations of ONE step of the adaptive Butcher-5 integrator (six right-hand
# sides), counted in the ISA of sbr_b5a's step loops (290 FMA x 2 + 169 MUL + 12 ADD + 6 RCP; with dosing 328 x 2 + 179 + 24 + 6);
# tests/test_isa_cpu.py asserts both.  How many steps an interval takes is decided per env (1, 2 or 4).
FP64_FLOP_PER_B5_STEP = {"plain": 767, "dosing": 865}
# vector float64 peak: 256 CUs x 4 SIMDs x 16 FMA lanes x 2 FLOP x 2.4 GHz = 78.6 TFLOP/s, i.e. half the 157.3 TFLOP/s float32
# vector figure of /opt/skills/guides/MI355X_MICROARCH.md (the guide lists no float64 vector row); one wave64 FMA = 4 cycles
FP64_VECTOR_PEAK_TFLOPS = 78.6
MAX_CLOCK_GHZ = 2.4                    # the same guide, chip-level parameters
SIMDS = 1024
PRIME_SECONDS = 0.3                    # untimed: brings the GPU to steady clocks before warm-up and timing
LARGE_LEG_ENVS = 262144                # the in-run large-batch leg: configs[3]'s total on ONE GPU (k_step's two-waves-per-SIMD build)


def episode_schedule(cfg=None):
    """Which kind of control interval every step() call of an episode runs: a list (one entry per call) of tuples of 0 (anoxic)
    / 1 (aerobic), from the reference's own logic - four sequential tests on the running time t, which advances by t_delta per
    interval in float64 (gym_SBR_oneshot.py:860, :896, :931, :963, :1122; the library's sbr_phase).  With the reference's
    constants: 463 calls, calls 0..50 anoxic, call 51 anoxic + aerobic (double step), 52..274 aerobic, 275 aerobic + anoxic,
    276..461 anoxic, 462 anoxic + aerobic = the done call.  `cfg` is anything with T_fill, T3_0, T3_end, T4_end, T5_end,
    t_delta (default: the reference's values)."""
    g = (lambda k, d: getattr(cfg, k, d)) if cfg is not None else (lambda k, d: d)
    t = g("T_fill", 0.021); t_delta = g("t_delta", 0.002 / 24 * 10)
    t30, t3e, t4e, t5e = g("T3_0", 0.06416666666666668), g("T3_end", 0.2516666666666667), g("T4_end", 0.4085000000000001), g("T5_end", 0.40933333333333344)
    calls = []
    while len(calls) < 100000:
        kinds = []
        if t < t30:
            kinds.append(0); t = t + t_delta
        if t30 <= t <= t3e:
            kinds.append(1); t = t + t_delta
        if t3e < t <= t4e:
            kinds.append(0); t = t + t_delta
        if t > t4e:
            kinds.append(1); t = t + t_delta
        calls.append(tuple(kinds))
        if t >= t5e:
            break
    return calls


def timed_window_start(steps, warmup, sched):
    """First timed call (0-based index into the episode) of a K-step region after W warm-up calls.  A region of less than an
    episode is centred on the first anoxic -> aerobic boundary so that its share of anoxic calls is the episode's; longer regions
    start at call W (they contain whole episodes)."""
    n = len(sched)
    if steps >= n:
        return warmup % n
    anoxic = [1.0 if k[-1] == 0 else 0.0 for k in sched]
    share = sum(anoxic) / n
    boundary = next((c for c, k in enumerate(sched) if len(k) > 1), None)       # the first double-step call
    if boundary is None:
        return warmup % n
    start = boundary - int(round(share * steps))
    return min(max(start, warmup), n - steps)
def reference_cpu():
    """The Python reference itself, timed by oracle/time_reference.py in the BUILD CONTAINER (the reference cannot travel to the
    GPU box) and committed as profiles/reference_cpu_timing.json: read here, never measured or imported by this file."""
    path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return {"value": None, "note": "profiles/reference_cpu_timing.json is missing: run oracle/time_reference.py in the build container"}
    return {"value": rec["value"], "unit": rec["unit"], "cores": rec["cores"], "value_all_cores": rec["value_all_cores"],
            "cores_all": rec["cores_all"], "hardware": rec["hardware"] + " (build container, NOT the GPU box)",
            "versions": rec["versions"], "what": rec["what"], "file": "profiles/reference_cpu_timing.json",
            "script": "oracle/time_reference.py"}


def walk_move(cur, pick, xp):
    """One move of the reference's action model (get_available_actions, gym_SBR_oneshot.py:440-459): `cur` [n, 2] set-points,
    `pick` [n, 2] integers in {0, 1, 2} choosing among the deltas (-0.1, 0, +0.1) / (-5, 0, +5); a move that would leave
    [0, 8] x [0, 15] is not available there, so the set-point stays (the draw is then among the available moves with the
    unavailable one's share going to `stay`).  xp = numpy or torch."""
    delta = (pick - 1) * (xp.asarray([0.1, 5.0]) if xp.__name__ == "numpy" else xp.tensor([0.1, 5.0], device=cur.device, dtype=cur.dtype))
    nxt = cur + delta
    hi = xp.asarray([8.0, 15.0]) if xp.__name__ == "numpy" else xp.tensor([8.0, 15.0], device=cur.device, dtype=cur.dtype)
    ok = (nxt >= 0) & (nxt <= hi)
    return xp.where(ok, nxt, cur)


def cpu_baseline(n_envs=16384, calls=463, policy="physical", scheme=1):
    """The CPU oracle (a C port of the same algorithm: the same integrator scheme, fp64, OpenMP over envs) timed on this box's
    host cores, on a bounded sample of the same workload.  Reported beside the GPU number; it is not the target.
    Its first pass also counts, per call of the episode, what the integrator did on this workload (the GPU does not report it):
    the share of 64-env groups (= wavefronts) with at least one lane dosing carbon, and under scheme 1 the mean step count per
    env and per wavefront (a wavefront runs its slowest lane's count)."""
    import numpy as np
    from oracle import sbr_oracle as O
    from gym_sbr2_amd.vec_env import load_influent_tables
    means, stds = load_influent_tables()
    cores = min(len(os.sched_getaffinity(0)), 16)
    physical = policy != "uniform"
    scen = ((4 + np.arange(n_envs) % 4) if physical else (np.arange(n_envs) % 8)).astype(np.int32)
    b = O.OracleBatch(n_envs, O.default_params(scheme=scheme), nthreads=cores)
    infl = b.mix(means, stds, scen, b.normals(0))
    rs = np.random.RandomState(0)
    if policy == "walk":
        acts, cur = [], np.column_stack([np.zeros(n_envs), np.full(n_envs, 15.0)])
        for _ in range(calls):
            cur = walk_move(cur, rs.randint(0, 3, (n_envs, 2)), np)
            acts.append(cur.astype(np.float32).astype(np.float64))
    else:
        acts = [np.column_stack([rs.uniform(0, 2.5 if physical else 8, n_envs), rs.uniform(0, 15, n_envs)]) for _ in range(calls)]
    best = 0.0
    per_call = {"dosing_wave_share": [], "steps_lane_mean": [], "steps_wave_mean": []}
    b.reset(infl)
    for a in acts:                     # untimed: the integrator's statistics per call (+ warms the caches)
        b.step(a, want_obs=False)
        ec = b.envs["ec_last"][:n_envs - n_envs % 64].reshape(-1, 64)
        per_call["dosing_wave_share"].append(float((ec != 0).any(axis=1).mean()))
        if scheme == 1:                # a phase-boundary call records its second interval's count; three calls per episode
            st = b.envs["scheme_steps"][:n_envs - n_envs % 64].reshape(-1, 64)
            per_call["steps_lane_mean"].append(float(st.mean())); per_call["steps_wave_mean"].append(float(st.max(axis=1).mean()))
    for _ in range(3):                 # best of three: shared hosts are noisy (about 0.5 s each: 7.6 M env-steps)
        b.reset(infl)
        b.step(acts[0], want_obs=False)
        t0 = time.perf_counter()
        for a in acts:
            b.step(a, want_obs=True)
        best = max(best, n_envs * calls / (time.perf_counter() - t0))
    # "1 core and all cores" - the same port on ONE thread, a sample sized for about a second
    n1 = 2048
    b1 = O.OracleBatch(n1, O.default_params(scheme=scheme), nthreads=1)
    infl1 = infl[:n1].copy()
    best1 = 0.0
    for _ in range(2):
        b1.reset(infl1)
        t0 = time.perf_counter()
        for a in acts:
            b1.step(a[:n1], want_obs=True)
        best1 = max(best1, n1 * calls / (time.perf_counter() - t0))
    return {"value": best, "unit": "env-steps/s", "cores": cores, "kind": "port", "scheme": scheme, "per_call": per_call,
            "sample": "%d envs x %d step() calls of the same workload, oracle/sbr_oracle.c (cfg.scheme = %d) with %d OpenMP threads, "
                      "best of 3" % (n_envs, calls, scheme, cores),
            "single_thread": {"value": best1, "unit": "env-steps/s", "cores": 1,
                              "sample": "%d envs x %d step() calls of the same workload on one thread, best of 2" % (n1, calls)},
            "reference": reference_cpu()}


def pmc_record(lib_hash, profiles_dir=None):
    """The committed PMC summary (scripts/pmc_summarise.py) that was measured on THE library being timed: the newest
    profiles/r*_pmc_traffic.json whose `library_source_hash` equals `lib_hash` (the content hash of sources + flags that
    gym_sbr2_amd.build writes next to the library it builds).  Returns (record, None) or (None, why not): a profile of another
    kernel must not be pasted into this run's line."""
    import glob
    paths = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "r*_pmc_traffic.json")), reverse=True)
    if not lib_hash:
        return None, "the loaded library carries no source hash (an A/B variant?): no committed PMC profile can be matched to it"
    seen = []
    for path in paths:
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("library_source_hash") == lib_hash:
            rec["_file"] = os.path.relpath(path, ROOT)
            return rec, None
        seen.append("%s: %s" % (os.path.basename(path), str(rec.get("library_source_hash"))[:12]))
    return None, ("no committed PMC profile was measured on this library (source hash %s; found %s): re-run scripts/profile_round.sh "
                  "and scripts/pmc_summarise.py" % (lib_hash[:12], "; ".join(seen) or "none"))


def hash_matched(pattern, lib_hash, profiles_dir=None, key="library_source_hash"):
    """Committed profiles/<pattern> records (newest first) whose `key` equals the hash of the library being timed."""
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), pattern)), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if lib_hash and rec.get(key) == lib_hash:
            rec["_file"] = os.path.relpath(path, ROOT)
            out.append(rec)
    return out


def larger_batches(lib_hash, profiles_dir=None):
    """north_star's '>= 40 % of HBM roofline on one MI355X' is a statement about the chip, and the configured
    65 536 envs are one wavefront per SIMD; what larger launches reach is carried by committed bench lines of THIS library
    (profiles/r*_bench_config2_n<envs>.json, `python bench.py --envs-per-gpu N`; each line records the hash of the library it
    timed as config.library_source_hash).  A committed constant under the same rule as `traffic`; absent if no file matches."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "r*_bench_config2_n*.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {})
        n = cfg.get("envs_per_gpu")
        if not lib_hash or cfg.get("library_source_hash") != lib_hash or n in (None, 65536) or str(n) in out:
            continue
        r = rec["roofline"]
        out[str(n)] = {"frac": r["frac"], "env_steps_per_s": rec["value"], "us_per_launch": rec["ms_per_step"] * 1e3,
                       "file": os.path.relpath(path, ROOT)}
        # the same round's rocprofv3 --kernel-trace --stats of that run, when scripts/profile_round.sh took one: the step kernel
        # alone, averaged over every launch of whole episodes (the wall-clock figure above also holds the resets)
        stats = path[:-len(".json")] + "_kernel_stats.csv"
        if os.path.exists(stats):
            import csv
            with open(stats) as f:
                for row in csv.DictReader(f):
                    if "k_step<" in row.get("Name", ""):
                        avg = float(row["AverageNs"])
                        out[str(n)].update({"kernel_trace_us_per_launch": avg * 1e-3, "kernel_trace_calls": int(row["Calls"]),
                                            "frac_kernel_trace": n * ALGO_BYTES_PER_ENV_STEP / (avg * 1e-9) / 1e9 / HBM_PEAK_GBPS,
                                            "kernel_trace_file": os.path.relpath(stats, ROOT)})
                        break
    return out or None


def serial_bound_record(lib_hash, profiles_dir=None):
    """The two measured constants of roofline.serial_bound (arithmetic of one call at the single-wave issue rate, period of an
    empty dependent launch) live in a committed profiles/r*_serial_bound.json keyed by the library they were measured on
    (they used to be literals in this file and went stale silently)."""
    recs = hash_matched("r*_serial_bound.json", lib_hash, profiles_dir)
    return recs[0] if recs else None


def loaded_library_hash():
    from gym_sbr2_amd import _capi
    try:
        with open(_capi.library_path() + ".srchash") as f:
            return f.read().strip() or None
    except OSError:
        return None


INTERVALS_PER_CYCLE = 528      # 24 + 48 + 223 + 186 + 11 + 36 control intervals (tests/golden/sbrv2_cycles.npz)
FILL_INTERVALS = 24            # the first phase integrates the filling right-hand side


def bench_cycle(args, torch, dist, world, rank, local_rank, dev, emit):
    """SBR-v2: every step is reset (influent draw) + one whole cycle, for all envs of this rank."""
    from gym_sbr2_amd import SbrEnv2Vec, _capi
    n_local = args.envs_per_gpu or 65536
    n_global = n_local * world
    cfg = _capi.default_config()
    if args.scheme is not None:
        cfg.scheme = args.scheme
    scheme = int(cfg.scheme)
    env = SbrEnv2Vec(n_local, device=local_rank, first_env_id=rank * n_local, config=cfg)
    scenario = ((torch.arange(n_local, device=dev) + rank * n_local) % 8).to(torch.int32)
    gen = torch.Generator(device=dev); gen.manual_seed(4321 + rank)
    pool = torch.rand(16, n_local, 3, device=dev, generator=gen)
    steps = max(1, min(args.steps, 64)) if args.steps == 1852 else args.steps       # default K is sized for config2

    def one(k):
        env.reset(seed=k, scenario=scenario)
        env.step(pool[k & 15], want_diag=False)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < PRIME_SECONDS:      # clock priming, see the module docstring
        one(50)
        torch.cuda.synchronize(dev)
    for k in range(max(1, min(args.warmup, 3))):
        one(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()                    # created and first used outside the timed region
    fence()
    t0 = time.perf_counter(); e0.record()
    for k in range(steps):
        one(100 + k)
    e1.record(); torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0          # clock read between the closing synchronise and the closing barrier, see main()
    fence()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
    per_launch_s = e0.elapsed_time(e1) * 1e-3 / steps
    achieved = n_local * INTERVALS_PER_CYCLE * ALGO_BYTES_PER_ENV_STEP / per_launch_s / 1e9
    # float64 work of one cycle, RK4 substep loops only (ISA counts, tests/test_isa_cpu.py): 24 filling intervals + 504 closed ones
    flop_per_cycle = SUBSTEPS * (FILL_INTERVALS * FP64_FLOP_PER_SUBSTEP["filling"]
                                 + (INTERVALS_PER_CYCLE - FILL_INTERVALS) * FP64_FLOP_PER_SUBSTEP["plain"])
    tflops = n_local * flop_per_cycle / per_launch_s / 1e12
    fp64 = {"achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
            "flop_per_env_step": flop_per_cycle / INTERVALS_PER_CYCLE,
            "note": "RK4 substep loops only, counted in the ISA (FMA = 2): a lower bound of the work"}
    if scheme == 1:       # the step count of every non-fill interval is decided per env; this path has no CPU sample to count them
        fp64 = {"achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "note": "cfg.scheme = 1: every one of the 528 intervals takes 1, 2 or 4 Butcher-5 steps (767 FLOP each) as each env's "
                        "state demands; read issue_slot_frac (committed PMC profile) for utilisation"}
    rec, why = pmc_record(loaded_library_hash())
    traffic = None
    if rec and "cycle" in rec and n_local == rec.get("envs_per_launch", 65536):
        waves = (n_local + 63) // 64
        cyc = rec["cycle"]
        traffic, why = cyc.get("hbm_bytes_per_launch"), "bytes per reset + cycle, a committed constant (%s, measured on this library)" % rec["_file"]
        if cyc.get("valu_insts_per_wave"):
            fp64["valu_insts_per_wave"] = cyc["valu_insts_per_wave"]
            fp64["issue_slot_frac"] = (cyc["valu_insts_per_wave"] * 4.0 * (waves / SIMDS if waves > SIMDS else 1.0)
                                       / (per_launch_s * MAX_CLOCK_GHZ * 1e9))
    out = {"metric": "env-steps/sec (batched)", "value": n_global * steps * INTERVALS_PER_CYCLE / elapsed, "unit": "env-steps/s",
           "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "SBR-v2 per-cycle env (SURVEY.md 8f-3): %d envs/GPU, one step = reset + one whole 12 h cycle = %d "
                                  "control intervals (cfg.scheme = %d); value counts control intervals" % (n_local, INTERVALS_PER_CYCLE, scheme),
                      "scheme": scheme, "envs_per_gpu": n_local, "envs_total": n_global, "cycles_per_s": n_global * steps / elapsed,
                      "clock_priming_s": PRIME_SECONDS,
                      "kernel": "k_cycle<float,float,%d> (+ k_cycle_reset)" % scheme},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "traffic": traffic, "traffic_unit": why, "avg_launch_us": per_launch_s * 1e6,
                        "fp64_valu": fp64, "headline": "fp64_valu",
                        "note": "FUSED kernel: `achieved`/`frac` are the 513-byte per-control-interval CONVENTION, not traffic - the kernel "
                                "moves one load and one store of the plant per cycle; it is bound by float64 VALU issue: read fp64_valu"}}
    env.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        emit(out)


class StepLeg:
    """One handle of n_local envs of the per-step workload (or the fused rollout), with everything a timed region needs: the
    resident action pool, the captured step graphs, episode bookkeeping and the event-timed launches.  The headline leg and the
    in-run large-batch leg are two instances of this."""

    def __init__(self, torch, dist, args, cfg, n_local, rank, world, dev_index, dist_up, fused=False, deterministic_influent=False):
        from gym_sbr2_amd import ShardedSbrOS, _capi
        self.torch, self.dist, self.args, self.capi = torch, dist, args, _capi
        self.n_local, self.world, self.rank, self.dist_up, self.fused = n_local, world, rank, dist_up, fused
        self.dev = torch.device("cuda", dev_index)
        self.n_global = n_local * world
        policy = args.policy
        self.physical = policy != "uniform"
        self.do_max = 2.5 if policy == "physical" else 8.0
        # the class the multi-GPU tests cover: contiguous shards by global env id, device = LOCAL_RANK
        self.sh = ShardedSbrOS(self.n_global, rank=rank, world=world, device=dev_index, out_dtype=torch.float32, config=cfg)
        self.env, first = self.sh.env, self.sh.start
        assert self.env.num_envs == n_local and self.env.device == self.dev, (self.env.num_envs, self.env.device, self.dev)
        gid = torch.arange(first, first + n_local, device=self.dev)
        self.scenario = ((4 + gid % 4) if self.physical else (gid % 8)).to(torch.int32)
        self.rnd0 = torch.zeros(n_local, 48, dtype=torch.float64, device=self.dev) if deterministic_influent else None
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(1234 + rank)
        if policy == "walk":           # history-dependent: one row per call of the episode, built once (untimed)
            rows, cur = [], torch.tensor([0.0, 15.0], device=self.dev).repeat(n_local, 1)
            for _ in range(CALLS_PER_EPISODE):
                pick = torch.randint(0, 3, (n_local, 2), device=self.dev, generator=gen)
                cur = walk_move(cur, pick, torch)
                rows.append(cur.clone())
            self.pool_rows = rows
        else:
            pool = torch.rand(64, n_local, 2, device=self.dev, generator=gen) * torch.tensor([self.do_max, 15.0], device=self.dev)
            self.pool_rows = [pool[k] for k in range(64)]      # the [N, 2] views, made once: indexing a tensor costs the host ~2 us per call
        self.n_rows = len(self.pool_rows)
        self.state = {"episode": 0, "in_episode": 0, "returns": None}
        self.seg_events, self.graphs = [], {}
        self.gbufs = self.sh.gather_buffers(torch.float32)       # float64 row, float32 send, float32 [n_global] recv: allocated once
        self.status_snap = torch.empty(n_local, dtype=torch.float64, device=self.dev)
        self.acct = {"end_of_episode_ms": 0.0, "reset_issue_ms": 0.0, "allgathers": 0, "call_ranges": []}
        self.event_pool = []

    # ---- the workload
    def reset(self):
        self.env.reset(seed=1000 + self.state["episode"], scenario=self.scenario, rnd=self.rnd0)
        self.state["episode"] += 1
        self.state["in_episode"] = 0

    def end_of_episode(self):
        # what the workload needs at an episode boundary, all asynchronous on the launch stream (no host sync, no allocation):
        # the per-env returns, collated over ranks by the one collective of the path (configs[3])
        self.state["returns"] = self.sh.gather_episode_returns_into(self.gbufs)      # all_gather_into_tensor when a group is up
        self.acct["allgathers"] += 1 if self.dist_up else 0
        self.env.ctrl_row(self.capi.C_STATUS, out=self.status_snap)     # snapshot only; reduced after the timed region

    # The step launches are issued as HIP-graph replays (chunks of <= 64 steps; sbr_step allocates nothing and synchronises
    # nothing, so a run of steps is capturable - DESIGN.md section 2): one host call per chunk instead of one ctypes call per
    # step.  With scheme 1 an anoxic call takes < 10 us on the GPU, and issuing 20 of them through Python took the host
    # 5 .. 13 us per step depending on the box (round 5: a driver-style run came out host-bound at 14.2 us per step on a slow
    # host, 11.1 on a fast one).  The captured launches are exactly the eager ones; --no-graphs issues them eagerly.
    @staticmethod
    def chunks(c0, m):
        while m > 0:
            r = c0 & 63
            ln = min(m, 64 - r)
            yield c0, ln
            c0 += ln; m -= ln

    def issue_steps(self, c0, m):
        for c, ln in self.chunks(c0, m):
            g = self.graphs.get((c % self.n_rows, ln))
            if g is not None:
                g.replay()
            else:
                for j in range(c, c + ln):
                    self.env.step(self.pool_rows[j % self.n_rows])

    def capture_for(self, schedule):
        """Capture (untimed; nothing executes during capture) a graph for every chunk the given (call index, count) segments
        will issue."""
        if self.args.no_graphs or self.fused:
            return
        for c0, m in schedule:
            for c, ln in self.chunks(c0, m):
                key = (c % self.n_rows, ln)
                if key not in self.graphs and ln > 1:
                    self.graphs[key] = self.env.capture_steps([self.pool_rows[j % self.n_rows] for j in range(c, c + ln)])

    def run(self, k_steps, record):
        done = 0
        st, acct = self.state, self.acct
        while done < k_steps:
            if st["in_episode"] == CALLS_PER_EPISODE:
                ta = time.perf_counter()
                self.end_of_episode()
                tb = time.perf_counter()
                self.reset()
                if record:
                    acct["end_of_episode_ms"] += (tb - ta) * 1e3
                    acct["reset_issue_ms"] += (time.perf_counter() - tb) * 1e3
            m = min(k_steps - done, CALLS_PER_EPISODE - st["in_episode"])
            if record:
                e0, e1 = self.event_pool.pop(), self.event_pool.pop()      # created before the timed region: no harness work in it
                e0.record()
            if self.fused:
                self.env.rollout(m, policy_seed=77)
            else:
                self.issue_steps(st["in_episode"], m)
            if record:
                e1.record()
                self.seg_events.append((e0, e1, m))
                acct["call_ranges"].append((st["in_episode"], st["in_episode"] + m))
            st["in_episode"] += m
            done += m

    def fence(self):
        torch = self.torch
        torch.cuda.synchronize(self.dev)
        if self.dist_up:
            self.dist.barrier()
        torch.cuda.synchronize(self.dev)

    def measure(self, steps, warmup, start_call):
        """Priming, warm-up and the timed region of `steps` calls whose first is call `start_call` of an episode.  Returns a dict
        of raw measurements (this rank's)."""
        torch = self.torch
        # untimed priming: one pass over everything an episode boundary touches (allocator growth, lazy loading of torch's
        # kernels, RCCL's first collective), then W warm-up steps.  Nothing here is counted.
        self.reset()
        self.end_of_episode()
        self.reset()
        # HIP events for the device time of the step launches.  torch creates an event at its first record(), and the first
        # timed record of a process costs 30-45 us on top (scripts/probes/sync_wait.py, profiles/r02_sync_wait.log): the warm-up
        # steps go through the same recording code path and every event of the pool is recorded once, so that the timed region
        # contains the workload only - with --steps 20 that harness cost was 12 % of the region.
        self.event_pool = [torch.cuda.Event(enable_timing=True)
                           for _ in range(2 * (steps // CALLS_PER_EPISODE + warmup // CALLS_PER_EPISODE + 6))]
        for ev in self.event_pool:
            ev.record()
        # every chunk the run will issue: whole episodes (priming, long regions), the untimed advance to the window, the warm-up's
        # W - 1 and 1 steps, the K timed steps (all modulo the episode length)
        lead = (start_call - warmup) % CALLS_PER_EPISODE
        sched, c = [(0, CALLS_PER_EPISODE)], 0
        for m_ in (lead, max(warmup - 1, 0), min(warmup, 1), steps):
            left = m_
            while left > 0:
                take = min(left, CALLS_PER_EPISODE - c)
                sched.append((c, take))
                c = (c + take) % CALLS_PER_EPISODE
                left -= take
        self.capture_for(sched)
        # The collector runs HERE, before the clocks are primed, and stays off until the timed region is over: a collection of a
        # torch process takes tens of milliseconds of host time during which the GPU idles and drops out of its steady clocks
        # (round 3, scripts/probes/rollout_sustained.py: the first 25 ms after such a gap run up to 24 % slower) - placed between
        # priming and timing, as it was, it undid the priming for every region shorter than ~30 ms.
        gc.collect(); gc.disable()
        if self.dist_up:                            # the first barrier of a process group is slow (lazy connection set-up): not
            self.fence()                            # between priming and timing either
        # clock priming, see the module docstring.  With a process group up every episode boundary issues the path's collective
        # (the return all-gather), so the NUMBER of priming episodes must be the same on every rank: a loop that each rank ends by
        # its own clock (as it was until round 6) lets ranks issue different numbers of collectives, after which the k-th
        # collective of one rank pairs with another rank's (k+1)-th - a different operation - and RCCL hangs.  Found by the four-rank
        # rehearsal of tests/test_gpu_parity.py; two ranks sharing one GPU had stayed in step by luck.  Every rank times its first
        # priming episode, the slowest rank's time (one all-reduce) fixes the count for all.
        t_prime = time.perf_counter()
        self.run(CALLS_PER_EPISODE, record=False)
        torch.cuda.synchronize(self.dev)
        n_more = None
        if self.dist_up:
            t_one = torch.tensor([time.perf_counter() - t_prime], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(t_one, op=self.dist.ReduceOp.MAX)
            n_more = max(0, int(PRIME_SECONDS / max(float(t_one.item()), 1e-4) + 0.5) - 1)
        while (n_more > 0) if n_more is not None else (time.perf_counter() - t_prime < PRIME_SECONDS):
            self.run(CALLS_PER_EPISODE, record=False)
            torch.cuda.synchronize(self.dev)
            if n_more is not None:
                n_more -= 1
        t_primed = time.perf_counter()
        if self.state["in_episode"] == CALLS_PER_EPISODE:         # the advance, warm-up and timing start at the first call of an episode
            self.end_of_episode()
            self.reset()
        self.run(lead, record=False)                              # untimed: up to the call the warm-up starts at
        self.run(max(warmup - 1, 0), record=True)
        # opening bracket: synchronise, barrier, the last warm-up step, synchronise (see the module docstring)
        torch.cuda.synchronize(self.dev)
        if self.dist_up:
            self.dist.barrier()
        self.run(min(warmup, 1), record=True)
        torch.cuda.synchronize(self.dev)
        self.seg_events.clear()
        self.acct.update(end_of_episode_ms=0.0, reset_issue_ms=0.0, allgathers=0, call_ranges=[])
        episodes_before = self.state["episode"]
        t0 = time.perf_counter()
        gap_ms = (t0 - t_primed) * 1e3              # (diagnostic) host time between the end of clock priming and the timed region
        self.run(steps, record=True)
        if self.dist_up and self.acct["allgathers"] == 0:     # no episode boundary fell into the K steps: the collective of the path
            self.end_of_episode()                             # still runs once inside the timed region of an N > 1 run (returns so far)
        t_issued = time.perf_counter()              # (diagnostic) the host has issued every launch of the region
        torch.cuda.synchronize(self.dev)            # closing bracket: synchronise, read the clock, then the barrier (+ synchronise)
        elapsed = time.perf_counter() - t0          # - the MAX over ranks (caller) is what makes it the time of the slowest rank, and
        self.fence()                                # a collective's own latency is not part of the K steps
        gc.enable()
        dev_ms = sum(a.elapsed_time(b) for a, b, _ in self.seg_events)
        launches = sum(m for _, _, m in self.seg_events) if not self.fused else len(self.seg_events)
        return {"elapsed": elapsed, "dev_ms": dev_ms, "launches": launches, "segments": len(self.seg_events),
                "host_issue_ms": (t_issued - t0) * 1e3, "gap_ms": gap_ms, "resets_timed": self.state["episode"] - episodes_before,
                "timed_calls": [c for lo, hi in self.acct["call_ranges"] for c in range(lo, hi)]}

    def device_plan_counts(self, calls, actions=None):
        """What cfg.scheme = 1 did on THIS leg's envs, counted on the device (round 6): one more episode of the same envs (the next
        reset seed, the same action rows and call order as the timed ones; untimed, eager), reading the plan row (SBR_C_PLAN: step
        count and slaved bit of each env's last interval) after every call of `calls`.  Returns per-env and per-wavefront means over
        those calls.  `actions` [calls, N, 2]: step through these rows instead of the pool (the fused rollout's own on-device draws,
        returned by sbr_rollout's actions_out: sbr_step with them passes through the rollout's states, to rounding)."""
        torch, capi = self.torch, self.capi
        if not hasattr(capi, "C_PLAN") or not calls:
            return None
        want = set(calls)
        row = torch.empty(self.n_local, dtype=torch.float64, device=self.dev)
        self.reset()
        lane_sum = wave_sum = slaved_sum = 0.0
        dose_sum = 0.0
        full = self.n_local - self.n_local % 64
        ec = torch.empty(self.n_local, dtype=torch.float64, device=self.dev)
        for c in range(max(want) + 1):
            self.env.step(self.pool_rows[c % self.n_rows] if actions is None else actions[c])
            if c in want:
                self.env.ctrl_row(capi.C_PLAN, out=row)
                p = row.to(torch.int64)
                n = (p & 127).to(torch.float64)
                lane_sum += float(n.mean().item())
                wave_sum += float(n[:full].view(-1, 64).max(dim=1).values.mean().item())
                slaved_sum += float(((p & 128) != 0).double().mean().item())
                self.env.ctrl_row(capi.C_EC_LAST, out=ec)
                dose_sum += float((ec[:full].view(-1, 64) != 0).any(dim=1).double().mean().item())
        self.state["in_episode"] = max(want) + 1
        k = float(len(want))
        return {"per_env_mean": lane_sum / k, "per_wavefront_mean": wave_sum / k, "slaved_share": slaved_sum / k,
                "dosing_wave_call_share": dose_sum / k, "calls_counted": len(want), "envs_counted": self.n_local,
                "source": ("device: SBR_C_PLAN / SBR_C_EC_LAST read after each of those calls in one more, untimed episode of the same envs "
                           "with the same action rows" if actions is None else
                           "device: SBR_C_PLAN / SBR_C_EC_LAST read after each call of one more, untimed episode stepped through sbr_step with "
                           "the actions the fused rollout's on-device policy draws (sbr_rollout's actions_out)")}

    def close(self):
        self.env.close()


def step_kernel_label(env, scheme, fused):
    """The kernel instantiation the library launches for this handle, from the library's own thresholds (sbr_query, round 6; until
    then bench.py repeated them as literals)."""
    from gym_sbr2_amd import _capi
    q = getattr(env, "query", None)
    if q is None:
        return None
    if fused:
        return "k_rollout<false,%d,%d>" % (scheme, q(_capi.Q_ROLLOUT_WAVES))
    return "k_step<float,float,%d,false,%d,%d>" % (q(_capi.Q_STEP_BLOCK), scheme, q(_capi.Q_STEP_WAVES))


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD process group through
    torch.distributed.run and relay rank 0's JSON line and the return code.  Runs before this process has imported torch or
    touched a GPU; a subprocess, never an exec."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, SBR_BENCH_SELF_LAUNCHED="1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return p.returncode if (p.returncode != 0 or lines) else 1


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1852)       # four episodes
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "config5", "cycle"])
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--policy", default="physical", choices=["physical", "uniform", "walk"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-leg", action="store_true", help="skip the in-run 262144-env leg (roofline.larger_batches['262144'])")
    ap.add_argument("--no-graphs", action="store_true", help="issue every step launch eagerly (one ctypes call per step)")
    ap.add_argument("--scheme", type=int, default=None, choices=[0, 1],
                    help="cfg.scheme: 1 (library default) adaptive Butcher-5 per interval, 0 ten RK4 substeps (rounds 1-4)")
    args = ap.parse_args(argv)
    if args.policy == "walk" and args.workload in ("config5", "cycle"):
        raise SystemExit("--policy walk is a per-step policy: the fused rollout draws its actions on the device, the per-cycle env takes three set-points per cycle")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))

    # Native libraries print to fd 1 (RCCL writes a five-line version banner when a communicator is created); the contract is
    # ONE JSON line on stdout, so fd 1 points at stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)

    import torch
    import torch.distributed as dist
    from gym_sbr2_amd import _capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    # One process per GPU: rank r drives device LOCAL_RANK.  SBR_BENCH_BACKEND=gloo is a REHEARSAL mode for a box with fewer
    # GPUs than ranks (tests/test_gpu_parity.py: two ranks on the one GPU of a test box): ranks then share devices
    # (LOCAL_RANK modulo the device count) and the collectives go through gloo, because RCCL refuses two ranks on one device.
    # It exercises everything an N > 1 run does except RCCL itself and is labelled in the JSON line; the driver never sets it.
    backend = os.environ.get("SBR_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("SBR_BENCH_BACKEND must be nccl (RCCL) or gloo (rehearsal)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dist = os.environ.get("SBR_BENCH_FORCE_DIST") == "1"      # rehearse the RCCL calls with a single rank
    if world > 1 or force_dist:
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)    # nccl == RCCL on ROCm

    if args.workload == "cycle":
        return bench_cycle(args, torch, dist, world, rank, dev_index, dev, emit)
    n_local = args.envs_per_gpu or (4096 if args.workload == "config1" else 65536)
    n_global = n_local * world
    physical = args.policy != "uniform"
    do_max = 2.5 if args.policy == "physical" else 8.0
    cfg = _capi.default_config()
    if args.scheme is not None:
        cfg.scheme = args.scheme
    elif args.workload == "config1":
        cfg.scheme = 0                    # BASELINE.json words configs[1] as "fixed-step RK4"; --scheme 1 runs the default scheme on it
    scheme = int(cfg.scheme)
    cfg.act_DO_max = do_max if args.policy != "walk" else 8.0      # what the fused rollout's on-device policy draws from (and clips to)
    fused = args.workload == "config5"
    dist_up = world > 1 or force_dist
    sched = episode_schedule(cfg)
    assert len(sched) == CALLS_PER_EPISODE, len(sched)
    anoxic_of = [1.0 if k[-1] == 0 else 0.0 for k in sched]
    start_call = timed_window_start(args.steps, args.warmup, sched) if not fused else args.warmup % CALLS_PER_EPISODE
    leg = StepLeg(torch, dist, args, cfg, n_local, rank, world, dev_index, dist_up, fused=fused,
                  deterministic_influent=args.workload == "config1")
    env = leg.env
    m = leg.measure(args.steps, args.warmup, start_call)
    elapsed = m["elapsed"]
    rank_elapsed, rank_devices = [elapsed], ["cuda:%d %s" % (dev_index, torch.cuda.get_device_name(dev))]
    if dist_up:
        # one small all-gather AFTER the timed region: every rank's own time and device, so that the line says whether RCCL saw
        # N ranks on N devices and how skewed they were; `value` uses the MAX
        ws = dist.get_world_size()
        mine = torch.tensor([elapsed, float(dev_index), float(rank)], dtype=torch.float64, device=dev)
        allr = torch.empty(ws * 3, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(ws, 3).cpu()
        rank_elapsed = [float(v) for v in allr[:, 0]]
        names = [None] * ws
        dist.all_gather_object(names, torch.cuda.get_device_name(dev))
        rank_devices = ["cuda:%d %s" % (int(allr[r, 1]), names[r]) for r in range(ws)]
        elapsed = max(rank_elapsed)

    resets_timed = m["resets_timed"]
    # dominant kernel: device time of the step launches of the timed region, from events on the launch stream
    dev_ms, launches = m["dev_ms"], m["launches"]
    per_launch_s = dev_ms * 1e-3 / max(launches, 1)
    calls_per_launch = 1 if not fused else args.steps / max(m["segments"], 1)
    achieved = n_local * calls_per_launch * ALGO_BYTES_PER_ENV_STEP / per_launch_s / 1e9
    timed_calls = m["timed_calls"]
    frac_anoxic = (sum(anoxic_of[c] for c in timed_calls) / len(timed_calls)) if timed_calls else None
    lib_hash = loaded_library_hash()

    # ---- the in-run large-batch leg: 262 144 envs on this one GPU, the same workload, bracket and K
    large = None
    if (world == 1 and not dist_up and args.workload == "config2" and args.envs_per_gpu is None and not args.no_large_leg
            and rank == 0):
        lleg = StepLeg(torch, dist, args, cfg, LARGE_LEG_ENVS, 0, 1, dev_index, False)
        lm = lleg.measure(args.steps, args.warmup, start_call)
        l_launch_s = lm["dev_ms"] * 1e-3 / max(lm["launches"], 1)
        l_step_s = lm["elapsed"] / args.steps
        l_bytes = LARGE_LEG_ENVS * ALGO_BYTES_PER_ENV_STEP
        l_counts = lleg.device_plan_counts(lm["timed_calls"][:CALLS_PER_EPISODE]) if scheme == 1 else None
        large = {"measured_in_this_run": True, "envs_per_launch": LARGE_LEG_ENVS, "steps": args.steps, "warmup": args.warmup,
                 "ms_per_step": l_step_s * 1e3, "env_steps_per_s": LARGE_LEG_ENVS * args.steps / lm["elapsed"],
                 "frac_wall": l_bytes / l_step_s / 1e9 / HBM_PEAK_GBPS, "frac": l_bytes / l_step_s / 1e9 / HBM_PEAK_GBPS,
                 "avg_launch_us": l_launch_s * 1e6, "frac_timed_launches": l_bytes / l_launch_s / 1e9 / HBM_PEAK_GBPS,
                 "launches_timed": lm["launches"], "resets_in_timed_region": lm["resets_timed"],
                 "timed_calls": [lm["timed_calls"][0], lm["timed_calls"][-1] + 1] if len(lm["timed_calls"]) <= CALLS_PER_EPISODE else "whole episodes",
                 "anoxic_share_of_timed_calls": sum(anoxic_of[c] for c in lm["timed_calls"]) / max(len(lm["timed_calls"]), 1),
                 "kernel": step_kernel_label(lleg.env, scheme, False), "b5_steps_per_interval": l_counts,
                 "algorithmic_bytes_per_launch": l_bytes,
                 "note": "a second handle of 262144 envs (configs[3]'s total on ONE GPU) timed in this very run with the headline's "
                         "bracket, priming, window and K; frac = frac_wall = 513 B x 262144 / ms_per_step / 8 TB/s (resets and episode "
                         "boundaries inside when K spans them)"}
        lleg.close()
        del lleg
        torch.cuda.empty_cache()

    # HBM bytes per launch and VALU instructions per wave from the PMC counters: collected offline with rocprofv3 --pmc
    # (scripts/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE / SQ passes, gfx950 fetch correction calibrated in the same
    # run) and committed under profiles/.  They are a COMMITTED CONSTANT, not a measurement of this run: attached only when the
    # profile was taken on the very library that is being timed (content hash of sources + flags), for the profiled batch size
    traffic, valu_per_wave = None, None
    rec, traffic_note = pmc_record(lib_hash)
    if rec and n_local != rec.get("envs_per_launch", 65536):
        rec, traffic_note = None, "the committed PMC profile is of %d envs per launch, this run has %d" % (rec.get("envs_per_launch", 65536), n_local)
    episode = None          # whole-episode launch time of the committed kernel trace of THIS library (hash-matched like traffic)
    if rec:                 # ... and of this cfg.scheme: the profile round runs the default one (k_step<float, float, 256, false, SCH, WAVES>)
        try:
            rec_scheme = int(rec["kernel_trace"]["kernel"].split("<")[1].split(">")[0].split(",")[4])
        except (KeyError, IndexError, ValueError):
            rec_scheme = 1
        if rec_scheme != scheme:
            rec, traffic_note = None, "the committed PMC profile is of cfg.scheme = %d, this run has %d" % (rec_scheme, scheme)
    if rec and rec.get("policy", "physical") != args.policy:
        rec, traffic_note = None, "the committed PMC profile is of --policy %s, this run has %s" % (rec.get("policy", "physical"), args.policy)
    if rec and not fused and args.workload == "config2":
        traffic = rec["hbm_bytes_per_launch"]
        valu_per_wave = rec.get("valu_insts_per_wave")
        episode = rec.get("kernel_trace")
        traffic_note = ("bytes per launch, a committed constant (%s, measured on this library: %.0f B per env-step vs %d algorithmic; "
                        "the internal layout also carries the Kla ring and bookkeeping rows, every byte moves once)"
                        % (rec["_file"], rec["hbm_bytes_per_env_step"], ALGO_BYTES_PER_ENV_STEP))
    elif rec and fused and "rollout" in rec:
        rr = rec["rollout"]
        traffic = rr["hbm_bytes_per_launch"]
        valu_per_wave = rr.get("valu_insts_per_wave")
        traffic_note = ("bytes per launch of %d calls, a committed constant (%s, measured on this library: %.2f B per env-step really "
                        "moved)" % (rr["calls_per_launch"], rec["_file"], rr["hbm_bytes_per_env_step"]))
    elif rec:
        traffic_note = "the committed PMC profile covers config2 and config5 only"
    # What cfg.scheme = 1 did on the timed envs, counted ON THE DEVICE (round 6; until then: the CPU oracle's sample): a replay of
    # the timed calls of one episode with the plan row read after every call.  After the timed region, untimed.
    counts = None
    if scheme == 1 and not fused and rank == 0:
        counts = leg.device_plan_counts(sorted(set(timed_calls))[:CALLS_PER_EPISODE])
    elif scheme == 1 and fused and rank == 0:
        # the fused kernel reports no plan (its state never leaves the registers): take the actions its policy draws for one more
        # episode and step through sbr_step with them - the same states to rounding, and the plan row says what each interval took
        leg.reset()
        _, racts = env.rollout(CALLS_PER_EPISODE, policy_seed=77, return_actions=True)
        counts = leg.device_plan_counts(list(range(CALLS_PER_EPISODE)), actions=racts)
        del racts
    # The CPU baseline (rank 0 of a 1-GPU run), after the timed region.  Its first pass also counts what the integrator did on
    # ITS sample of the workload (other random draws): kept as a cross-check of the device's count.
    cpu = cpu_baseline(policy=args.policy, scheme=scheme) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    def over_timed_calls(key):
        v = cpu["per_call"].get(key) if cpu else None
        return (sum(v[c] for c in timed_calls) / len(timed_calls)) if (v and timed_calls) else None
    dosing_share = counts["dosing_wave_call_share"] if counts else over_timed_calls("dosing_wave_share")
    waves = (n_local + 63) // 64
    if scheme == 1:
        # scheme 1: the work per interval is decided per env (1, 2 or 4 Butcher-5 steps of 767 FLOP; with dosing 865); a
        # wavefront executes its slowest lane's count with the other lanes masked.  `achieved` counts the USEFUL work (the mean
        # count per env at the closed-reactor figure: a lower bound), `executed_flop_per_env_step` what the wavefronts issued.
        steps_lane = counts["per_env_mean"] if counts else over_timed_calls("steps_lane_mean")
        steps_wave = counts["per_wavefront_mean"] if counts else over_timed_calls("steps_wave_mean")
        flop_per_step = steps_lane * FP64_FLOP_PER_B5_STEP["plain"] if steps_lane else None
        tflops = n_local * calls_per_launch * flop_per_step / per_launch_s / 1e12 if flop_per_step else None
        fp64 = {"achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / FP64_VECTOR_PEAK_TFLOPS if tflops else None, "flop_per_env_step": flop_per_step,
                "executed_flop_per_env_step": steps_wave * FP64_FLOP_PER_B5_STEP["plain"] if steps_wave else None,
                "b5_steps_per_interval": {"per_env_mean": steps_lane, "per_wavefront_mean": steps_wave,
                                          "slaved_share": counts["slaved_share"] if counts else None,
                                          "source": counts["source"] if counts else ("CPU oracle's sample of the same workload (other "
                                                                                     "random draws), same calls" if cpu else None),
                                          "cpu_oracle_sample": {"per_env_mean": over_timed_calls("steps_lane_mean"),
                                                                "per_wavefront_mean": over_timed_calls("steps_wave_mean")} if cpu else None,
                                          "flop_per_step": FP64_FLOP_PER_B5_STEP, "rk4_equivalent_flop": SUBSTEPS * FP64_FLOP_PER_SUBSTEP["plain"]},
                "anoxic_share_of_timed_calls": frac_anoxic,
                "note": "cfg.scheme = 1: Butcher-5 step loops only, counted in the ISA (FMA = 2); the step counts are the DEVICE's own "
                        "(the plan row of the timed envs, read after each of the timed calls in one more, untimed episode of the same envs); "
                        "scheme 0 spent 4300 FLOP per env-step on the same intervals"}
    else:
        # scheme 0: the RK4 substep loops only, ALL counted at the closed-reactor loop's 426 FLOP per substep - a lower bound.  A
        # wave with at least one lane dosing carbon runs the dosing loop instead (464 FLOP per substep); how many do is a property
        # of the policy, not of the phase (config.dosing_wave_call_share).
        flop_per_step = SUBSTEPS * FP64_FLOP_PER_SUBSTEP["plain"]
        tflops = n_local * calls_per_launch * flop_per_step / per_launch_s / 1e12
        fp64 = {"achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                "flop_per_env_step": flop_per_step, "anoxic_share_of_timed_calls": frac_anoxic,
                "note": "RK4 substep loops only, every call counted at the closed-reactor loop's ISA count (FMA = 2): a lower bound of "
                        "the work (waves with a dosing lane run 464 instead of 426 FLOP per substep); one wave per SIMD issues a "
                        "v_fma_f64 every 5.2 cycles and v_mul/v_add_f64 every 4.3 (scripts/probes/fp64_issue.hip), so ~0.8 of the "
                        "nominal peak is what a single resident wave can reach"}
    if valu_per_wave:
        # share of the VALU issue slots of the launch that carried an instruction: instructions per wave (per launch: one call, or
        # the 463 calls of a fused launch) x 4 cycles (one wave64 fp64 instruction at the nominal rate) x waves per SIMD /
        # (launch time x max clock)
        scale = calls_per_launch / rec["rollout"]["calls_per_launch"] if fused else 1.0
        fp64["issue_slot_frac"] = (valu_per_wave * scale * 4.0 * (waves / SIMDS if waves > SIMDS else 1.0)
                                   / (per_launch_s * MAX_CLOCK_GHZ * 1e9))
        fp64["valu_insts_per_wave"] = valu_per_wave * scale
    integ = "adaptive Butcher-5 (cfg.scheme = 1: 1, 2 or 4 steps per interval and env)" if scheme == 1 else "fixed-step RK4 (10 substeps)"
    workload = {"config1": "configs[1]: 4096 envs/GPU, %s, deterministic influent, per-step API" % integ,
                "config2": "configs[2]: 65536 envs/GPU, stochastic influent perturbations, %s, per-step API" % integ,
                "config5": "configs[4]: 65536 envs/GPU, fused on-GPU random-policy rollout, %s" % integ}[args.workload]
    if n_local not in (4096, 65536):
        workload = workload.replace("65536 envs/GPU", "%d envs/GPU" % n_local).replace("4096 envs/GPU", "%d envs/GPU" % n_local)
    if world > 1:
        rel = ("= configs[3]'s 262144" if n_global == 262144 else
               "= %.3g x configs[3]'s 262144 (weak scaling keeps the 1-GPU line's per-GPU batch)" % (n_global / 262144.0))
        workload += ("; configs[3] shape: sharded over %d GPUs by global env id (gym_sbr2_amd.ShardedSbrOS), one RCCL all-gather of the "
                     "episode returns per episode boundary, at least one inside the timed region (config.allgathers_in_timed_region); "
                     "envs_total %d %s" % (world, n_global, rel))
    # ---- roofline of the dominant kernel.  Three figures exist for the per-step path; `frac` is the CONSERVATIVE one (VERDICT r4
    # item 2): the whole-episode average of the committed rocprofv3 --kernel-trace --stats run of THIS library (hash-matched)
    # when there is one, and never above what this run's own wall clock allows (513 B x N / ms_per_step).
    #   frac_timed_launches  513 B x N / mean device time of the launches this run timed (HIP events on the launch stream)
    #   frac_wall            513 B x N / (wall time of the K steps / K): includes resets, episode boundaries and host gaps
    #   frac_episode         513 B x N / AVERAGE duration of every k_step launch of the committed kernel trace of whole episodes
    algo_bytes = n_local * calls_per_launch * ALGO_BYTES_PER_ENV_STEP
    frac_timed = achieved / HBM_PEAK_GBPS
    frac_wall = n_local * ALGO_BYTES_PER_ENV_STEP / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS      # a "step" = one call per env
    frac_episode = (n_local * ALGO_BYTES_PER_ENV_STEP / (episode["average_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS) if episode else None
    frac = min(frac_episode, frac_wall) if frac_episode is not None else frac_wall
    sb = serial_bound_record(lib_hash) if (not fused and n_local == 65536 and traffic) else None
    serial_bound = None
    if sb:
        mem_us = traffic / (HBM_PEAK_GBPS * 1e3)
        arith = sb["arithmetic_us"]
        n_anoxic = int(sum(anoxic_of))
        arith_mean = (n_anoxic * arith["anoxic"] + (CALLS_PER_EPISODE - n_anoxic) * arith["aerobic"]) / float(CALLS_PER_EPISODE)
        total = mem_us + arith_mean + sb["dependent_launch_floor_us"]
        serial_bound = {"memory_us": mem_us, "arithmetic_us": arith, "arithmetic_us_episode_mean": arith_mean,
                        "anoxic_calls_per_episode": n_anoxic, "aerobic_calls_per_episode": CALLS_PER_EPISODE - n_anoxic,
                        "dependent_launch_floor_us": sb["dependent_launch_floor_us"], "sum_us": total,
                        "frac_at_bound": n_local * ALGO_BYTES_PER_ENV_STEP / (total * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                        "note": "what three measured numbers add up to when nothing overlaps (one wave per SIMD): memory_us from `traffic` "
                                "of this library at the roofline's own rate; arithmetic (PIDs + integration, per-wave median of the stamp "
                                "build) and the period of an empty dependent launch are committed measurements of this library (%s)"
                                % sb["_file"]}
    lb = None
    if world == 1 and args.envs_per_gpu is None and args.workload == "config2":
        lb = {}
        committed = (larger_batches(lib_hash) or {}) if (args.policy == "physical" and args.scheme is None) else {}    # they are lines of the default workload
        for k_, v_ in committed.items():                               # committed constants of this library, labelled as such
            v_ = dict(v_, measured_in_this_run=False, committed_constant=True)
            lb[k_ if (k_ != str(LARGE_LEG_ENVS) or large is None) else k_ + "_committed_record"] = v_
        if large is not None:
            lb[str(LARGE_LEG_ENVS)] = large
        lb = lb or None
    roofline = {"bound": "hbm", "achieved": frac * HBM_PEAK_GBPS, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac,
                "frac_is": ("frac_episode" if (frac_episode is not None and frac_episode <= frac_wall) else "frac_wall"),
                "frac_timed_launches": frac_timed, "frac_wall": frac_wall, "frac_episode": frac_episode,
                "traffic": traffic, "traffic_unit": traffic_note,
                "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                "avg_launch_us": per_launch_s * 1e6, "launches_timed": launches,
                "avg_launch_us_episode": episode["average_ns"] * 1e-3 if episode else None,
                "frac_episode_source": ("%s: %d k_step launches, average %.0f ns (committed constant, measured on this library)"
                                        % (episode["file"], episode["calls"], episode["average_ns"])) if episode else
                                       "no committed kernel trace of this library and batch size",
                "larger_batches": lb,
                "timed_region_ms": {"wall": elapsed * 1e3, "step_kernels_device": dev_ms, "host_issue": m["host_issue_ms"],
                                    "host_in_end_of_episode": leg.acct["end_of_episode_ms"],
                                    "host_in_reset_issue": leg.acct["reset_issue_ms"], "since_clock_priming": m["gap_ms"]},
                "fp64_valu": fp64, "serial_bound": serial_bound,
                "headline": "fp64_valu" if fused else "hbm",
                "note": ("FUSED kernel: plant and controllers stay in registers for the whole launch, so `achieved`/`frac` are the "
                         "513-byte per-step CONVENTION, not traffic (`traffic` is what really moves); the kernel is bound by float64 "
                         "VALU issue - read fp64_valu (issue_slot_frac = share of the VALU issue slots used, frac = FLOP share of the "
                         "78.6 TFLOP/s vector peak)") if fused else
                        ("the prescribed roofline is HBM (513 algorithmic bytes per env-step, SURVEY.md 8d); the kernel's actual "
                         "bound is float64 VALU issue plus the kernel boundary (DESIGN.md section 5), reported in fp64_valu")}
    st_bits = leg.status_snap.to(torch.int64)
    few = len(timed_calls) <= CALLS_PER_EPISODE
    out = {
        "metric": "env-steps/sec (batched)",
        "value": n_global * args.steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "envs_per_gpu": n_local, "envs_total": n_global, "calls_per_episode": CALLS_PER_EPISODE,
                   "resets_in_timed_region": resets_timed, "clock_priming_s": PRIME_SECONDS,
                   "timed_calls": ([timed_calls[0], timed_calls[-1] + 1] if (few and timed_calls) else
                                   "whole episodes: %d calls from call %d on" % (len(timed_calls), timed_calls[0] if timed_calls else 0)),
                   "timed_window": ("calls [first, last + 1) of an episode (0-based); a region shorter than an episode straddles the "
                                    "first anoxic -> aerobic phase boundary (call 51, a double step) with the episode's own share of "
                                    "anoxic calls; the episode is advanced untimed to call first - warmup"),
                   "anoxic_share_of_timed_calls": frac_anoxic, "anoxic_share_of_an_episode": sum(anoxic_of) / len(anoxic_of),
                   "collective_backend": ("none" if not dist_up else "nccl (RCCL)" if backend == "nccl" else
                                          "gloo - REHEARSAL: %d ranks share %d GPU(s), not a scaling measurement" % (world, ndev)),
                   "ranks": dist.get_world_size() if dist_up else 1,
                   "backend_reported": dist.get_backend() if dist_up else None,
                   "rank_elapsed_ms": [t * 1e3 for t in rank_elapsed],
                   "rank_skew_ms": (max(rank_elapsed) - min(rank_elapsed)) * 1e3,
                   "rank_devices": rank_devices,
                   "self_launched": os.environ.get("SBR_BENCH_SELF_LAUNCHED") == "1",
                   "allgathers_in_timed_region": leg.acct["allgathers"],
                   "allgather_bytes_per_rank": 4 * n_local if dist_up else 0,
                   "opening_bracket": "synchronize, barrier, last warm-up step, synchronize",
                   "policy": args.policy,
                   "step_issue": ("eager: one ctypes call per step" if (args.no_graphs or fused) else
                                  "HIP-graph replays of <= 64 captured sbr_step launches (%d graphs)" % len(leg.graphs)),
                   "actions": (("the reference's own action model (get_available_actions, gym_SBR_oneshot.py:440-459): from [0, 15], each call "
                                "moves each set-point by one of (-0.1, 0, +0.1) / (-5, 0, +5) inside [0, 8] x [0, 15]; float32, resident in HBM; "
                                "influent scenarios 4..7") if args.policy == "walk" else
                               ("per-call random set-points u_DO ~ U[0, %.1f], u_EC ~ U[0, 15], float32, resident in HBM; influent "
                                "scenarios %s" % (do_max, "4..7 (4 + global id mod 4)" if physical else "0..7 (global id mod 8)"))),
                   "scheme": scheme, "library_source_hash": lib_hash,
                   "dosing_wave_call_share": dosing_share,
                   "kernel": step_kernel_label(env, scheme, fused)},
        "roofline": roofline,
        "env_status": {"near_pole_frac_last_episode": float(((st_bits & _capi.ST_NEAR_POLE) != 0).float().mean().item())
                       if leg.state["episode"] > 2 else None,
                       "negative_frac_last_episode": float(((st_bits & _capi.ST_NEGATIVE) != 0).float().mean().item())
                       if leg.state["episode"] > 2 else None,
                       "nonfinite": int(((st_bits & _capi.ST_NONFINITE) != 0).sum().item()),
                       "note": "sticky per-env flags of the last finished episode (SBR_ST_* in include/sbr_amd.h): the physical policy "
                               "keeps every env inside the model's domain; the uniform one drives ammonia negative in most envs (the "
                               "reference model has no guards).  Under cfg.scheme = 1 the cost of a call depends on the states (step "
                               "counts are chosen per env): see fp64_valu.b5_steps_per_interval"},
    }
    if cpu is not None:
        cpu = dict(cpu)
        cpu.pop("per_call", None)             # 3 x 463 floats: summarised above (fp64_valu, config.dosing_wave_call_share)
        out["cpu_baseline"] = cpu
    env.close()
    if dist_up:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
