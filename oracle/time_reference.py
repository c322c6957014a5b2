#!/usr/bin/env python3
"""Times the UNMODIFIED Python reference on this machine's host cores (TEST / MEASUREMENT INFRASTRUCTURE; build container
only - /root/reference does not exist on the GPU box, and this script is never imported by the product, the tests or
bench.py: bench.py only READS the JSON it writes).

What is timed: `SbrOS` (`SBROS-v1`, gym_SBR/envs/gym_SBR_oneshot.py:843 `step`), whole 463-call episodes with the constant
action [2.0, 5.0] after `np.random.seed(k); env.reset()` - the reference's own control flow, SciPy's LSODA, stdout suppressed
(the env prints).  One env per process (the reference keeps its state in module globals), imported behind the same `gym`
stand-in oracle/gen_golden.py uses.  Two figures, as SURVEY.md 8(d) asks: one process on one core, and P processes on P
cores (default: every core the container has) started together, throughput = all env-steps / the time until the last
process has finished.  reset() is inside the timed region of an episode and is not counted as steps, like in bench.py.

    python oracle/time_reference.py [--episodes E] [--procs P] [--out profiles/reference_cpu_timing.json]
"""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CALLS = 463


def _worker(rank, episodes, barrier, out):
    """`episodes` whole episodes in this process; reports (steps, seconds, return of the first episode)."""
    sys.path.insert(0, HERE)
    import numpy as np
    import gen_golden as GG                      # only its gym stand-in and import helper: nothing of the reference is copied
    M = GG.import_reference()
    env = M.SbrOS()
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(0)
        env.reset()
        env.step([2.0, 5.0])                     # first call of the process untimed (imports, first LSODA call)
    barrier.wait()
    t0 = time.perf_counter()
    steps, first_return = 0, None
    for ep in range(episodes):
        with contextlib.redirect_stdout(io.StringIO()):
            np.random.seed(ep)
            env.reset()
            total, done = 0.0, False
            while not done:
                _, _, r, done, _ = env.step([2.0, 5.0])
                total += r
                steps += 1
        if first_return is None:
            first_return = total
    out.put((rank, steps, time.perf_counter() - t0, first_return))


def run(procs, episodes):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs + 1), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, episodes, barrier, q)) for r in range(procs)]
    for p in ps:
        p.start()
    barrier.wait()                               # every process has imported the reference and made its first call
    t0 = time.perf_counter()
    res = [q.get() for _ in ps]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join()
    steps = sum(r[1] for r in res)
    return {"processes": procs, "episodes_per_process": episodes, "env_steps": steps, "wall_s": wall,
            "env_steps_per_s": steps / wall, "per_process_env_steps_per_s": [r[1] / r[2] for r in sorted(res)],
            "episode_return_seed0": sorted(res)[0][3]}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu_timing.json"))
    args = ap.parse_args()
    import numpy
    import scipy
    one = run(1, args.episodes)
    many = run(args.procs, args.episodes)
    assert abs(one["episode_return_seed0"] / -0.8789670883455737 - 1) < 1e-12, one["episode_return_seed0"]   # SURVEY.md 8c anchor
    rec = {
        "what": "the unmodified Python reference (gym_SBR SbrOS, SBROS-v1; SciPy LSODA), whole 463-call episodes, constant action "
                "[2.0, 5.0], one env per process; measured by oracle/time_reference.py in the BUILD CONTAINER (not the GPU box)",
        "unit": "env-steps/s",
        "one_process": one, "all_cores": many,
        "value": one["env_steps_per_s"], "cores": 1,
        "value_all_cores": many["env_steps_per_s"], "cores_all": args.procs,
        "hardware": "%s, %d logical cores visible to the container" % (cpu_model(), len(os.sched_getaffinity(0))),
        "versions": {"python": platform.python_version(), "numpy": numpy.__version__, "scipy": scipy.__version__},
        "calls_per_episode": CALLS,
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("reference: %.0f env-steps/s on one core, %.0f on %d processes (%s)" % (rec["value"], rec["value_all_cores"], args.procs, rec["hardware"]))


if __name__ == "__main__":
    main()
