"""Host side of the dense trajectory export (gym_sbr2_amd/envs/sbr_os.py::_hermite), no GPU: the interpolant that carries the
device's RK4 nodes and node slopes onto the reference's output grids."""
import numpy as np

from gym_sbr2_amd.envs.sbr_os import _hermite


def _nodes(f, df, span, s_n, ncomp=14):
    t = np.linspace(0.0, span, s_n + 1)
    return np.stack([f(t + 0.1 * j) for j in range(ncomp)], 1), np.stack([df(t + 0.1 * j) for j in range(ncomp)], 1)


def test_quintic_hermite_reproduces_polynomials_up_to_degree_five_and_the_nodes():
    span, s_n = 8.3e-4, 10
    c = np.array([0.3, -1.2, 0.7, 2.1, -0.4, 0.9])
    s = 1.0 / span                                                     # keep the powers O(1)
    f = lambda t: sum(ck * (t * s) ** k for k, ck in enumerate(c))     # noqa: E731
    df = lambda t: sum(k * ck * (t * s) ** (k - 1) * s for k, ck in enumerate(c) if k)   # noqa: E731
    nodes, slopes = _nodes(f, df, span, s_n)
    tau = np.linspace(0.0, span, 57)
    got = _hermite(nodes, slopes, span, tau)
    ref = np.stack([f(tau + 0.1 * j) for j in range(14)], 1)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    at_nodes = _hermite(nodes, slopes, span, np.linspace(0.0, span, s_n + 1))
    assert np.abs(at_nodes - nodes).max() < 1e-13 * np.abs(nodes).max()


def test_error_falls_as_h_to_the_sixth_and_two_nodes_fall_back_to_the_cubic():
    span = 1.0
    f, df = (lambda t: np.sin(3 * t)), (lambda t: 3 * np.cos(3 * t))
    tau = np.linspace(0.0, span, 201)
    errs = []
    for s_n in (8, 16, 32):
        nodes, slopes = _nodes(f, df, span, s_n, ncomp=1)
        errs.append(np.abs(_hermite(nodes, slopes, span, tau)[:, 0] - f(tau)).max())
    assert errs[0] / errs[1] > 40 and errs[1] / errs[2] > 40           # 2^6 = 64 in the limit
    nodes, slopes = _nodes(f, df, 0.05, 1, ncomp=1)                     # a span with a single substep: cubic through both ends
    mid = _hermite(nodes, slopes, 0.05, np.array([0.0, 0.025, 0.05]))[:, 0]
    assert abs(mid[0] - f(0.0)) < 1e-15 and abs(mid[2] - f(0.05)) < 1e-15 and abs(mid[1] - f(0.025)) < 1e-7


def test_numpy_rng_draw_follows_the_reference_call_pattern():
    """reset() of the reference-shaped classes takes the influent noise from np.random.randn(48) exactly as buffer_tank() does:
    every scenario block but 0 draws twice and uses the second vector (buffer_tank3.py:206/:224 ... :971/:989), scenario 0 once
    (:68).  The fixtures hold the vector the reference actually used after np.random.seed(seed)."""
    import numpy as np
    from conftest import golden
    from gym_sbr2_amd.envs.sbr_os import reference_randn
    e = golden("sbros_const_2_5")
    np.random.seed(int(e["seed"]))
    assert np.array_equal(reference_randn(6), e["rnd"])                 # SbrOS.reset: scenario 6, second draw
    g = golden("sbrv2_cycles")
    np.random.seed(11)                                                  # oracle/gen_golden.py run_cycle_env(seed=11)
    for c in range(len(g["rnd"])):
        assert np.array_equal(reference_randn(0), g["rnd"][c])          # SbrEnv2.reset: scenario 0, one draw per reset
