"""are two dependent chains of half-batches, each on its OWN raw hipStream_t, faster per 65536-env step than
one chain of full batches?  Everything goes through the raw C ABI with arguments converted once (no torch.cuda.stream context,
no per-call Python objects): the host costs ~3.5 us per sbr_step call, so two calls per step stay below the kernel time.

  one     1 handle x 65536 envs, one stream                               (the baseline: 14.84 us per launch by rocprof in round 3)
  two     2 handles x 32768 envs (first_env_id 0 / 32768), two streams, stepped alternately
  four    4 handles x 16384 envs, four streams
  graph2  the two chains captured fork/join into ONE HIP graph of 64 steps, replayed
  two256  like `two`, for a library built with -DSBR_SMALL_BATCH=16384 (256-thread workgroups at 32768 envs)

usage: python scripts/probes/two_groups_raw.py [modes ...] [--steps K] [--reps R] [--total N]
Prints the period per step of ALL envs (wall clock between two device synchronisations, best and median of R repetitions,
each after an untimed warm-up inside the same episode) and the implied env-steps/s.
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gym_sbr2_amd import SbrOSVec, _capi  # noqa: E402


def make_groups(groups, total, dev):
    n = total // groups
    out = []
    for g in range(groups):
        st = torch.cuda.Stream(device=dev)
        env = SbrOSVec(n, first_env_id=g * n)
        gid = torch.arange(g * n, (g + 1) * n, device=dev)
        scen = (4 + gid % 4).to(torch.int32)
        gen = torch.Generator(device=dev); gen.manual_seed(1234 + g)
        pool = torch.rand(64, n, 2, device=dev, generator=gen) * torch.tensor([2.5, 15.0], device=dev)
        args = [(C.c_void_p(pool[j].data_ptr()), C.c_void_p(env.obs.data_ptr()), C.c_void_p(env.state.data_ptr()),
                 C.c_void_p(env.reward.data_ptr()), C.c_void_p(env.done.data_ptr())) for j in range(64)]
        out.append(dict(env=env, stream=st, raw=C.c_void_p(st.cuda_stream), scen=scen, pool=pool, args=args, h=env._h))
    return out


def reset_all(gs, seed):
    for g in gs:
        with torch.cuda.stream(g["stream"]):
            g["env"].reset(seed=seed, scenario=g["scen"])
    torch.cuda.synchronize()


def run_eager(gs, steps, reps, label, total):
    f = gs[0]["env"].lib.sbr_step
    per = []
    # steady clocks first (the chip needs ~25 ms of sustained work)
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < 0.3:
        reset_all(gs, 1)
        for s in range(400):
            for g in gs:
                a, o, st_, r, d = g["args"][s & 63]
                f(g["h"], a, o, st_, r, d, g["raw"])
        torch.cuda.synchronize()
    for rep in range(reps):
        reset_all(gs, 10 + rep)
        for s in range(20):                                # calls 0..19 untimed
            for g in gs:
                a, o, st_, r, d = g["args"][s & 63]
                f(g["h"], a, o, st_, r, d, g["raw"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(20, 20 + steps):
            for g in gs:
                a, o, st_, r, d = g["args"][s & 63]
                f(g["h"], a, o, st_, r, d, g["raw"])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per.append(((t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6))
    per.sort()
    best, med = per[0][0], per[len(per) // 2][0]
    print("%-8s %d group(s) x %6d envs: period per %d-env step best %.2f us, median %.2f us (host issue %.2f us) = %.3fe9 env-steps/s"
          % (label, len(gs), total // len(gs), total, best, med, per[0][1], total / best / 1e3), flush=True)
    return best


def run_graph2(gs, steps, reps, total, per_graph=64):
    """Both chains in one graph: capture on stream A, fork to B, 64 x (A: step group 0, B: step group 1), join."""
    f = gs[0]["env"].lib.sbr_step
    sa, sb = gs[0]["stream"], gs[1]["stream"]
    graph = torch.cuda.CUDAGraph()
    reset_all(gs, 1)
    for s in range(3):
        for g in gs:
            a, o, st_, r, d = g["args"][s & 63]
            f(g["h"], a, o, st_, r, d, g["raw"])
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        with torch.cuda.graph(graph, stream=sa):
            sb.wait_stream(sa)                                  # fork
            for s in range(per_graph):
                for g in gs:
                    a, o, st_, r, d = g["args"][s & 63]
                    f(g["h"], a, o, st_, r, d, g["raw"])
            sa.wait_stream(sb)                                  # join
    torch.cuda.synchronize()
    n_rep = max(1, steps // per_graph)
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < 0.3:
        reset_all(gs, 1)
        with torch.cuda.stream(sa):
            for _ in range(6):
                graph.replay()
        torch.cuda.synchronize()
    per = []
    for rep in range(reps):
        reset_all(gs, 10 + rep)
        with torch.cuda.stream(sa):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(sa):
            for _ in range(n_rep):
                graph.replay()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / (n_rep * per_graph) * 1e6)
    per.sort()
    print("%-8s 2 groups x %6d envs, one graph of %d steps: period per %d-env step best %.2f us, median %.2f us = %.3fe9 env-steps/s"
          % ("graph2", total // 2, per_graph, total, per[0], per[len(per) // 2], total / per[0] / 1e3), flush=True)


def main():
    argv = sys.argv[1:]
    def opt(name, default):
        if name in argv:
            i = argv.index(name); v = int(argv[i + 1]); del argv[i:i + 2]; return v
        return default
    steps, reps, total = opt("--steps", 384), opt("--reps", 7), opt("--total", 65536)
    modes = argv or ["one", "two", "four", "graph2", "one"]
    dev = torch.device("cuda", 0)
    print("library:", _capi.library_path(), "| steps", steps, "reps", reps, "total envs", total, flush=True)
    for m in modes:
        ng = {"one": 1, "two": 2, "two256": 2, "four": 4, "graph2": 2, "eight": 8}[m]
        gs = make_groups(ng, total, dev)
        if m == "graph2":
            run_graph2(gs, steps, reps, total)
        else:
            run_eager(gs, steps, reps, m, total)
        for g in gs:
            g["env"].close()


if __name__ == "__main__":
    main()
