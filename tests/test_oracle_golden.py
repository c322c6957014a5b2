"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference.

What is asserted, and why it is worded this way (measured numbers in DESIGN.md, section Parity):

* Layer 1 (oracle/sbr_ref.py, same LSODA as the reference) reproduces every golden episode -
  intervals per call, done, Kla, EC, rewards, observations, terminal phases - to ~1e-11 relative.
* Layer 2 (oracle/sbr_oracle.c, fixed-step RK4 = what the HIP kernels compute) is bit-identical to
  layer 1 run in RK4 mode.
* RK4 vs the reference, OPEN loop: from the golden start state of each of the 466 intervals of every
  episode, with the golden Kla/EC, the end state is inside the gate (|d| <= 1e-5|ref| + 1e-5 scale).
* RK4 vs the reference, CLOSED loop (chained over the whole episode): inside the gate for four of
  the six episodes.  In `random_b` and `zeros` the golden trajectory ITSELF is 2.6x / 1.35x outside
  the gate (in Ss only) relative to the reference's OWN run at tight tolerance (1e-12): the
  reference's default LSODA tolerance (1.5e-8) is amplified by the NO3-PID -> carbon-dosing loop
  (dEC = 100 dSno, dSs ~ 3030 dEC per interval).  No integrator other than the bit-identical LSODA
  can follow it there.  The closed-loop bar is therefore asserted on ALL SIX episodes against
  fixtures of the unmodified reference run with odeint forced to rtol = atol = 1e-12
  (sbros_*_tight.npz); the default-tolerance exceedance is printed, not asserted.
"""
import numpy as np
import pytest
from conftest import BENCH_SCENARIOS, EPISODES, HELDOUT_EPISODES, SCENARIO_EPISODES, gate, golden, valid_calls

from oracle import sbr_oracle as O
from oracle import sbr_params as P
from oracle import sbr_ref as R

CLOSED_LOOP_OK = ["const_2_5", "random_a", "max", "det_influent"]
CLOSED_LOOP_REFERENCE_NOISE = ["random_b", "zeros"]
ALL_EPISODES = EPISODES + SCENARIO_EPISODES        # the 24 episodes the plan thresholds of cfg.scheme = 1 were fitted on (round 5)
WITH_HELDOUT = ALL_EPISODES + HELDOUT_EPISODES     # + the ten held-out ones of round 6: identity tests run on all 34


def _scn(e):
    return int(e["scenario"]) if "scenario" in e.files else 6      # the six tight fixtures of round 3 are scenario 6


def test_constants_match_reference():
    c = golden("constants")
    for key, val in [("T1_end", P.T1_END), ("T3_0", P.T3_0), ("T3_end", P.T3_END), ("T4_end", P.T4_END),
                     ("T5_end", P.T5_END), ("So_sat", P.SO_SAT), ("dt", P.DT), ("t_delta", P.T_DELTA),
                     ("EC_conc", P.EC_CONC)]:
        assert float(c[key]) == val, key
    assert [p[2] for p in P.phase_times()] == c["lens"].tolist()
    p = O.default_params()
    for key, val in [("T1_end", p.T_fill), ("T3_0", p.T3_0), ("T3_end", p.T3_end), ("T4_end", p.T4_end),
                     ("T5_end", p.T5_end), ("So_sat", p.So_sat), ("dt", p.dt), ("t_delta", p.t_delta)]:
        assert float(c[key]) == val, key


def test_rhs_known_answers():
    k = golden("rhs_kat")
    n = len(k["X"])
    zero = np.zeros(n)
    # reaction and idle right-hand sides: bit-exact in both oracle layers
    assert np.array_equal(O.eval_rhs(0, k["X"], k["kla"], k["ec"]), k["d_reaction"])
    assert np.array_equal(O.eval_rhs(2, k["X"], k["kla"], zero), k["d_idle"])
    for i in range(n):
        assert np.array_equal(R.rhs_reaction(k["X"][i], 0.0, k["kla"][i], k["ec"][i]), k["d_reaction"][i])
        assert np.array_equal(R.rhs_idle(k["X"][i], 0.0, k["kla"][i]), k["d_idle"][i])
    # filling: the reference's in-place dilution block (gym_SBR_oneshot.py:1523-1551) is (x*V)/V at
    # EC = 0 - not a bitwise no-op.  Not restated => <= 1 ulp of the largest term, i.e. 1e-14 relative.
    ref = k["d_filling"]
    tol = 1e-14 * np.abs(ref).max(axis=1, keepdims=True)
    c_fill = O.eval_rhs(1, k["X"], k["kla"], zero, k["loading"])
    assert np.all(np.abs(c_fill - ref) <= tol)
    # ... and that really is the whole difference: with the round trip put back it is bit-exact
    for i in range(n):
        x, ld = k["X"][i], k["loading"][i]
        r = R.conversion(x, k["kla"][i])
        y = np.array([x[0]] + [x[j] * x[0] / (x[0] + 0.0) for j in range(1, 14)])
        d = np.array([ld[0]] + [r[j] + (ld[0] / x[0]) * (ld[j] - y[j]) for j in range(1, 14)])
        assert np.array_equal(d, ref[i])
        assert np.array_equal(R.rhs_fill(x, 0.0, k["kla"][i], ld), c_fill[i])       # layers agree bitwise
    # anchor from SURVEY.md section 8c: RHS at x0_init, Kla = 100, EC = 0
    d = k["d_reaction"][12]
    assert abs(d[2] + 1.3417542) < 1e-6 and abs(d[5] + 228.94854228) < 1e-6


def test_influent_mix_exact(tables):
    means, stds = tables
    ik = golden("influent_kat")
    b = O.OracleBatch(len(ik["scenario"]))
    assert np.array_equal(b.mix(means, stds, ik["scenario"], ik["rnd"]), ik["mixed"])
    for s, r, m in zip(ik["scenario"], ik["rnd"], ik["mixed"]):
        assert np.array_equal(R.influent_mix(means[s], stds[s], r), m)


def test_interval_row_count_is_9_or_10():
    """SURVEY.md section 8a assumed 9 rows always; the reference produces 9 or 10 (fp rounding of
    (t+t_delta)-t), which changes how far back the reward looks."""
    e = golden("sbros_const_2_5")
    n = e["iv_n_rows"]
    assert set(n.tolist()) == {9, 10} and (n == 10).sum() == 266
    calc = [int(((t + P.T_DELTA) - t) / P.DT) for t in e["iv_t_start"]]
    assert calc == n.tolist()


@pytest.mark.parametrize("name", WITH_HELDOUT)
def test_layer1_lsoda_restatement_is_bit_identical_to_reference(name, tables):
    """Same LSODA, restated algorithm: every state, reward, observation and controller output of every
    call equals the reference's BIT FOR BIT (no tolerance), terminal phases included - on all eight influent scenarios
    (round 5), and also where an episode has left the model's domain (same LSODA, same inputs, same garbage)."""
    e = golden("sbros_" + name)
    assert int(e["crashed"]) == 0 and int(e["n_calls"]) == 463
    env = R.SbrOsRef(tables)
    obs = env.reset(rnd=e["rnd"], scenario=_scn(e))
    assert env.n_fill_rows == int(e["n_fill_rows"]) == 252
    assert np.array_equal(env.influent, e["influent_mixed"])
    assert np.array_equal(env.x, e["x_postfill"])
    assert np.array_equal(obs[0], e["reset_obs_DO"]) and np.array_equal(obs[1], e["reset_obs_EC"])
    assert np.array_equal(env.kla_hist[-9:], e["reset_Kla_tail"])
    assert env.ie_do == float(e["reset_ie_DO"]) and env.ie_ec == float(e["reset_ie_EC"])
    rewards = []
    for k in range(int(e["n_calls"])):
        obs, state, r, done, _ = env.step(e["actions"][k])
        assert done == bool(e["step_done"][k]) and len(env.intervals) == e["step_n_intervals"][k]
        assert env.last["n_rows"] == e["iv_n_rows"][np.where(e["iv_call"] == k)[0][-1]]
        assert np.array_equal(env.x, e["step_x_end"][k]), k
        assert env.kla_last == e["step_Kla"][k] and env.ec_last == e["step_EC"][k]
        assert env.ie_do == e["step_ie_DO"][k] or done      # the idle phase updates ie_DO on the done call
        assert env.ie_ec == e["step_ie_EC"][k]
        assert r == e["step_reward"][k]
        assert np.array_equal(obs[0], e["step_obs_DO"][k]) and np.array_equal(obs[1], e["step_obs_EC"][k])
        assert np.array_equal(state, e["step_state"][k])
        rewards.append(r)
    assert float(np.sum(rewards)) == float(e["episode_return"])
    assert env.qw == float(e["term_Qw"])
    assert np.array_equal(env.x_after_draw, e["term_x_after_draw"])
    assert np.array_equal(env.x_after_idle, e["term_x_after_idle"])
    assert env.kla_idle == float(e["term_Kla_idle"]) and env.n_idle_rows == int(e["term_n_idle_rows"])


def test_closed_loop_sensitivity_is_why_two_episodes_leave_the_gate(tables):
    """Measured, not assumed: the SAME code (layer 1, LSODA) restarted from a post-fill state that
    differs by 1 ulp in Sno drifts by < 1e-7 of the gate in `const_2_5` but ~1e3 times more in `zeros`
    (amplification ~1e7 from the NO3-PID -> dosing loop).  LSODA's default local error (1.5e-8) times
    that amplification is what puts the reference's own `zeros`/`random_b` trajectories outside 1e-5."""
    drift = {}
    for name in ("const_2_5", "zeros"):
        e = golden("sbros_" + name)
        runs = []
        for kick in (False, True):
            env = R.SbrOsRef(tables)
            env.reset(rnd=e["rnd"])
            if kick:
                env.x = env.x.copy()
                env.x[9] = np.nextafter(env.x[9], np.inf)
            xs = []
            for k in range(int(e["n_calls"]) - 1):
                env.step(e["actions"][k])
                xs.append(env.x.copy())
            runs.append(np.array(xs))
        drift[name] = gate(runs[1], runs[0]).max()
    assert drift["const_2_5"] < 1e-7
    assert drift["zeros"] > 100 * drift["const_2_5"]


@pytest.mark.parametrize("name", EPISODES + ["scn4_phys", "scn5_c25", "scn7_phys", "scn0_phys"])
def test_pid_known_answers_open_loop(name, tables):
    """Both PIDs + phase logic, open loop: controller memory and plant state of call k-1 are injected
    from the fixture, one step() is run, and Kla, EC and both integrals must equal the reference's
    values of call k to rounding (they are computed before the integration, so no integrator noise).
    Covers anti-windup on both clamps, the integrators that wind while their actuator is forced to 0,
    the previous-Kla bias and the phase-boundary double steps."""
    e = golden("sbros_" + name)
    n = int(e["n_calls"])
    py = R.SbrOsRef(tables, integrator="rk4", scheme=0)
    py.reset(rnd=e["rnd"], scenario=_scn(e))
    b = O.OracleBatch(1, O.default_params(scheme=0))
    b.reset(e["influent_mixed"][None])
    checked = clamped_hi = clamped_lo = doubles = 0
    for k in range(1, n):
        ivs = np.where(e["iv_call"] == k)[0]
        j = k - 1
        # EC of the interval before call k's first interval, and the Kla history, from the interval log
        i0 = ivs[0]
        py.t, py.x = float(e["step_t"][j]), e["step_x_end"][j].copy()
        py.so_m1, py.so_m2 = float(e["step_So_m1"][j]), float(e["step_So_m2"][j])
        py.sno_m1, py.sno_m2 = float(e["step_Sno_m1"][j]), float(e["step_Sno_m2"][j])
        py.ie_do, py.ie_ec = float(e["step_ie_DO"][j]), float(e["step_ie_EC"][j])
        py.kla_last, py.ec_last = float(e["step_Kla"][j]), float(e["step_EC"][j])
        hist = ([0.0] * 10 + e["iv_Kla"][:i0].tolist())[-10:]
        py.kla_hist = hist
        py.ec_prev = float(e["iv_EC"][i0 - 2]) if i0 >= 2 else 0.0
        env = b.envs[0]
        env["t"], env["x"] = py.t, py.x
        env["so_m1"], env["so_m2"], env["sno_m1"], env["sno_m2"] = py.so_m1, py.so_m2, py.sno_m1, py.sno_m2
        env["ie_do"], env["ie_ec"], env["kla_last"], env["ec_last"] = py.ie_do, py.ie_ec, py.kla_last, py.ec_last
        env["kla_hist"], env["ec_prev"], env["done"] = hist, py.ec_prev, 0.0
        a = e["actions"][k]
        py.step(a)
        b.step(a[None])
        first = py.intervals[0]
        assert len(py.intervals) == len(ivs) == b.envs["n_intervals"][0]
        # first interval of the call: controller outputs depend only on injected memory
        for got_kla, got_ec in ((first["kla"], first["ec"]),):
            assert abs(got_kla - e["iv_Kla"][i0]) <= 1e-13 * max(1.0, abs(e["iv_Kla"][i0]))
            assert abs(got_ec - e["iv_EC"][i0]) <= 1e-13 * 5e-4
        if len(ivs) == 1:
            for got, ref in ((py.ie_do, e["step_ie_DO"][k]), (py.ie_ec, e["step_ie_EC"][k]),
                             (b.envs["ie_do"][0], e["step_ie_DO"][k]), (b.envs["ie_ec"][0], e["step_ie_EC"][k]),
                             (b.envs["kla_last"][0], e["step_Kla"][k])):
                assert abs(got - ref) <= 1e-13 * max(1e-3, abs(ref))
            assert abs(b.envs["ec_last"][0] - e["step_EC"][k]) <= 1e-13 * 5e-4
        else:
            doubles += 1
        clamped_hi += int(e["iv_Kla"][i0] == 240.0) + int(e["iv_EC"][i0] == 5e-4)
        clamped_lo += int(e["iv_kind"][i0] == 1 and e["iv_Kla"][i0] == 0.0)
        checked += 1
    assert checked == n - 1 and doubles == 3
    if name in ("const_2_5", "max"):
        assert clamped_hi > 0                      # the upper clamps (with anti-windup) are exercised


@pytest.mark.parametrize("name", WITH_HELDOUT)
def test_layer2_c_rk4_is_bit_identical_to_layer1_rk4(name, tables):
    """Scheme 0 (RK4 x 10 per interval); scheme 1 has its own test below."""
    means, stds = tables
    e = golden("sbros_" + name)
    b = O.OracleBatch(1, O.default_params(scheme=0))
    cobs = b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
    py = R.SbrOsRef(tables, integrator="rk4", scheme=0)
    pobs = py.reset(rnd=e["rnd"], scenario=_scn(e))
    assert np.array_equal(cobs[0], np.r_[pobs[0], pobs[1]]) and np.array_equal(b.envs["x"][0], py.x)
    n = int(e["n_calls"])
    for k in range(n):
        o, s, r, d = b.step(e["actions"][k][None])
        po, ps, pr, pd, _ = py.step(e["actions"][k])
        assert np.array_equal(o[0], np.r_[po[0], po[1]]) and np.array_equal(s[0], ps)
        assert abs(r[0] - pr) <= 4e-18 and bool(d[0]) == pd
        if k < n - 1:                      # on the done call the C env holds the post-idle state
            assert np.array_equal(b.envs["x"][0], py.x)
    assert np.array_equal(b.envs["x"][0], py.x_after_idle)
    assert b.envs["qw"][0] == py.qw


@pytest.mark.parametrize("name", ALL_EPISODES)
def test_rk4_open_loop_every_interval_inside_gate(name):
    """'The reference odeint step on identical initial states': from the reference's own state at the start of each interval,
    with its Kla / EC, RK4 (10 substeps) ends inside the gate of the reference's end state - every interval of every episode
    on all eight influent scenarios whose start state is not within 50 % of a Monod pole (valid_calls: scenarios 0..3 get
    there around call 290 under either policy, scenario 5 under constant [2, 5]; nothing on the bench's workload does)."""
    e = golden("sbros_" + name)
    nv = valid_calls(e)
    worst, n_checked = 0.0, 0
    for i in range(len(e["iv_kind"])):
        if e["iv_call"][i] > nv:           # the interval STARTS from the end state of call iv_call - 1
            continue
        span = e["iv_t_end"][i] - e["iv_t_start"][i]
        x1 = O.rk4(0, e["iv_x_start"][i], span, 10, e["iv_Kla"][i], e["iv_EC"][i])
        worst = max(worst, gate(x1, e["iv_x_end"][i]).max())
        n_checked += 1
    assert worst <= 1.0, worst          # measured worst: 0.51 (random_b), 0.34 on the scenario fixtures (scn1_phys, So, call 51)
    assert n_checked == 466 if nv == 463 else n_checked >= 286
    if name.endswith("_phys") and _scn(e) in BENCH_SCENARIOS:
        assert nv == 463 and int(e["domain_exit_call"]) == -1      # the bench's workload stays inside the model's domain


def _c_episode(e, tables, scheme=0):
    means, stds = tables
    b = O.OracleBatch(1, O.default_params(scheme=scheme))
    b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
    xs = []
    for k in range(int(e["n_calls"])):
        b.step(e["actions"][k][None])
        xs.append(b.envs["x"][0].copy())
    return np.array(xs), b


@pytest.mark.parametrize("name", CLOSED_LOOP_OK)
def test_rk4_closed_loop_inside_gate_of_reference(name, tables):
    e = golden("sbros_" + name)
    xs, b = _c_episode(e, tables)
    n = int(e["n_calls"])
    assert gate(xs[:n - 1], e["step_x_end"][:n - 1]).max() <= 1.0
    assert gate(xs[n - 1], e["term_x_after_idle"]).max() <= 1.0
    assert abs(b.envs["ret"][0] - float(e["episode_return"])) < 1e-5 * abs(float(e["episode_return"]))
    assert abs(b.envs["qw"][0] / float(e["term_Qw"]) - 1) < 1e-5


@pytest.mark.parametrize("name", ALL_EPISODES)
def test_rk4_closed_loop_inside_gate_of_the_reference_at_tight_tolerance(name, tables):
    """The closed-loop bar, against the reference itself: tests/golden/sbros_*_tight.npz are the episodes run by the unmodified
    reference with every odeint call forced to rtol = atol = 1e-12 (oracle/gen_golden.py tight_episodes / scenario_episodes).
    RK4 with 10 substeps stays inside the 1e-5 gate over the whole chained episode, terminal phases, return and wastage
    included (measured worst: 0.51, So in random_b) - on all eight influent scenarios, up to the call at which an episode comes
    within 50 % of a Monod pole (valid_calls), beyond which no two float64 computations of this model agree."""
    e = golden("sbros_%s_tight" % name)
    xs, b = _c_episode(e, tables)
    n, nv = int(e["n_calls"]), valid_calls(e)
    assert float(e["odeint_tol"]) == 1e-12 and n == 463
    assert gate(xs[:min(nv, n - 1)], e["step_x_end"][:min(nv, n - 1)]).max() <= 1.0
    assert np.array_equal(b.envs["t"], e["step_t"][-1:])             # same time recurrence, same phase switches
    if nv == n:
        assert gate(xs[n - 1], e["term_x_after_idle"]).max() <= 1.0
        assert abs(b.envs["ret"][0] / float(e["episode_return"]) - 1) < 1e-5
        assert abs(b.envs["qw"][0] / float(e["term_Qw"]) - 1) < 1e-5


@pytest.mark.parametrize("name", CLOSED_LOOP_REFERENCE_NOISE)
def test_rk4_closed_loop_where_reference_noise_exceeds_gate(name, tables, capsys):
    """Informational: on these two episodes the reference's DEFAULT-tolerance trajectory is itself outside the gate of its
    own tight-tolerance run (in Ss only; LSODA's 1.5e-8 local error amplified by the NO3-PID -> dosing loop), so nothing
    but the bit-identical LSODA run can follow it to 1e-5.  The bar is the test above; this one only keeps the numbers
    visible and checks that the excess is confined to Ss."""
    e, tight = golden("sbros_" + name), golden("sbros_%s_tight" % name)
    n = int(e["n_calls"])
    xs, _ = _c_episode(e, tables)
    g_gold = gate(e["step_x_end"][:n - 1], tight["step_x_end"][:n - 1])
    g_rk4 = gate(xs[:n - 1], e["step_x_end"][:n - 1])
    with capsys.disabled():
        print("\n[info] %s: reference(default tol) vs reference(1e-12): %.3f of the gate; RK4 vs reference(default tol): %.3f"
              % (name, g_gold.max(), g_rk4.max()))
    others = [i for i in range(14) if i != 2]
    assert g_rk4[:, others].max() <= 1.0 and g_gold[:, others].max() <= 1.0    # every component but Ss is inside
    # ... and the excess in Ss is the reference's own, not the integrator's: RK4's distance to the default-tolerance run may not
    # exceed 3 gates nor differ by more than 5 % from the distance of the reference's own tight run to it (measured: random_b
    # 2.57 vs 2.58, zeros 1.35 vs 1.35) - a regression that grew the gap would not pass
    assert 1.0 < g_gold.max() < 3.0 and g_rk4.max() < 3.0
    assert abs(g_rk4.max() / g_gold.max() - 1.0) < 0.05, (g_rk4.max(), g_gold.max())


def test_philox_normals_are_standard():
    b = O.OracleBatch(256)
    z = b.normals(seed=0)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03 and np.isfinite(z).all()
    assert not np.array_equal(z[0], z[1])
    assert np.array_equal(O.OracleBatch(1, first_env_id=5).normals(0)[0], z[5])   # keyed by GLOBAL env id


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="control build needs a CPU with FMA")
def test_control_rounding_noise_is_amplified_to_gate_level_by_the_closed_loop(tables):
    """CONTROL for the GPU tolerances.  The SAME C source built without and with FMA contraction, same 1024
    random-action episodes: the two differ only in rounding (1e-16 per operation), yet the closed loop amplifies
    that to ~1e-9 of the gate in the median and to ~1e-3..1e-1 of the gate in the worst env/call (measured on 4096
    envs: median 2e-11, p99 7e-9, worst transient 0.17).  Hence free-running GPU-vs-oracle comparisons over many
    random episodes are asserted statistically (median, p99, max <= gate) and the tight comparison is done with
    the oracle re-synchronised to the device before every call (tests/test_gpu_parity.py)."""
    means, stds = tables
    n, ncall = 1024, 463
    scen = (np.arange(n) % 8).astype(np.int32)
    strict = O.OracleBatch(n, O.default_params(scheme=0), nthreads=4)
    infl = strict.mix(means, stds, scen, np.zeros((n, 48)))
    rs = np.random.RandomState(2)
    acts = [np.column_stack([rs.uniform(0, 8, n), rs.uniform(0, 15, n)]).astype(np.float32).astype(np.float64)
            for _ in range(ncall)]
    runs = []
    for variant in ("", "_fma"):
        prev = O.use_variant(variant)
        try:
            b = O.OracleBatch(n, O.default_params(scheme=0), nthreads=4)
            b.reset(infl)
            xs, dones = [], []
            for c in range(ncall - 1):
                _, _, _, d = b.step(acts[c], want_obs=False)
                xs.append(b.envs["x"].copy()); dones.append(d.copy())
            runs.append((np.array(xs), np.array(dones), b.envs["ie_ec"].copy()))
        finally:
            O.use_variant(prev)
    g = gate(runs[1][0], runs[0][0]).max(axis=2)          # [call][env]
    assert np.array_equal(runs[0][1], runs[1][1])
    assert np.median(g) < 1e-7 and np.percentile(g, 99) < 1e-4 and g.max() <= 1.0
    assert g.max() > 1e-6                                  # i.e. >= 1e10 ulp: the amplification is real
    assert np.abs(runs[0][2] - runs[1][2]).max() < 1e-9    # no clamp/anti-windup branch flipped (that would be ~1e-4)


def test_g2anet_reward_known_answers():
    """cfg.reward_kind = 1: module_reward_continuous_G2ANET.py, values from the reference function itself on 96 states
    that straddle every kink (Ss = 0, 10; So = 1.5; Sno, Snh = 4)."""
    import ctypes as C
    k = golden("reward_g2anet_kat")
    fn = O.lib().sbro_reward_g2anet
    fn.restype = C.c_double
    got = np.array([fn(O._p(np.ascontiguousarray(x))) for x in k["X"]])
    assert np.array_equal(got, k["reward"])
    assert len(set(np.round(k["reward"], 6))) > 40


def test_oci_reward_known_answers():
    """cfg.reward_kind = 2: module_reward_continuous.py:4-65 (reward of SbrEnv3/SbrEnv4), values from the reference
    function itself on 120 ragged Kla lists: all three batch_type branches, ammonia either side of the 4 g/m3 penalty."""
    import ctypes as C
    k = golden("reward_oci_kat")
    fn = O.lib().sbro_reward_oci
    fn.restype = C.c_double
    fn.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]
    got = np.array([fn(k["so_sat"][i], k["kla_last"][i], k["kla_sum"][i], int(k["batch_type"][i]), k["qin"][i],
                       k["qw"][i], k["q_eff"][i], k["snh_eff"][i]) for i in range(len(k["reward"]))])
    assert np.array_equal(got, k["reward"])
    assert set(k["batch_type"].tolist()) == {0, 1, 2} and (k["reward"] < -200).sum() >= 5      # penalty branch taken


def test_oci_reward_episode_bookkeeping():
    """reward_kind = 2 inside the SBROS-v1 plant: every ordinary call pays the reaction-interval branch on the Kla it
    applied; the done call pays the end-of-cycle branch on sum(Kla) of the episode's whole list - the 252 reset entries
    [0, k_fill]*126 (gym_SBR_oneshot.py:323), one Kla per interval (466) and the idle phase's.  The list is rebuilt
    here from the Kla history after every call and summed with python's own sum()."""
    infl = golden("sbros_const_2_5")["influent_mixed"]
    p = O.default_params(); p.reward_kind = 2
    b = O.OracleBatch(1, p)
    q = O.default_params(); q.terminal = 0                   # same plant (the reward does not feed back), stops before settling
    c = O.OracleBatch(1, q)
    b.reset(infl); c.reset(infl)
    so_sat, td = 8.000000000006622, 0.002 / 24
    klas = [0, float(b.envs["kla_last"][0])] * 126
    a = np.array([[2.0, 5.0]])
    for call in range(463):
        _, _, r, d = b.step(a)
        c.step(a)
        n_iv = int(c.envs["n_intervals"][0])
        klas += c.envs["kla_hist"][0][-n_iv:].tolist()
        if not d[0]:
            assert r[0] == 0.5 - so_sat / (1.8 * 1000) * (1.32 * klas[-1] * td)
            assert b.envs["kla_sum"][0] == sum(klas)
    assert d[0] and len(klas) == 252 + 466
    klas.append(float(b.envs["kla_hist"][0][-1]))            # Sim_idle's append (:2578)
    assert b.envs["kla_sum"][0] == sum(klas)
    qw, snh = float(b.envs["qw"][0]), float(c.envs["x"][0][10])
    want = 0.5 - (so_sat / (1.8 * 1000) * (1.32 * sum(klas) * td) + (0.05 * qw + 0.004 * 0.66)) + (0 if snh < 4 else -246)
    assert r[0] == want and 0.0 < want < 0.5 and sum(klas) > 1000.0


def test_substep_count_is_set_by_accuracy_and_stability():
    """Why cfg.substeps = 10.  Open loop over ALL golden intervals (2796), C oracle, gate against the reference's
    own LSODA end state: 8 substeps of classical RK4 miss the 1e-5 gate, 10 meet it with a factor ~2 in hand, and the error
    falls as h^4 (so it is truncation error, not the reference's).  Stability: h = dt keeps lambda*h of the stiffest mode seen on
    the golden states near 1, well inside RK4's real stability interval of 2.785."""
    import ctypes as C
    lib, p = O.lib(), O.default_params()
    lib.sbro_rk4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_double, C.c_double,
                             C.POINTER(C.c_double)]
    lib.sbro_rhs_reaction.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
    worst = {n: 0.0 for n in (6, 8, 10, 20)}
    count, lam_max = 0, 0.0
    for name in EPISODES:
        e = golden("sbros_" + name)
        for i in range(len(e["iv_kind"])):
            span = float(e["iv_t_end"][i]) - float(e["iv_t_start"][i])
            kla, ec = float(e["iv_Kla"][i]), float(e["iv_EC"][i])
            for n in worst:
                x = e["iv_x_start"][i].copy()
                lib.sbro_rk4(C.byref(p), 0, O._p(x), span, n, kla, ec, None)
                worst[n] = max(worst[n], gate(x, e["iv_x_end"][i]).max())
            count += 1
            if i % 8 == 0:          # the stiffest direction is dissolved oxygen: d(dSo/dt)/dSo by central differences
                x = e["iv_x_start"][i].copy()
                d = 1e-6
                xp, xm, fp, fm = x.copy(), x.copy(), np.empty(14), np.empty(14)
                xp[8] += d; xm[8] -= d
                lib.sbro_rhs_reaction(C.byref(p), O._p(xp), kla, ec, O._p(fp))
                lib.sbro_rhs_reaction(C.byref(p), O._p(xm), kla, ec, O._p(fm))
                lam_max = max(lam_max, abs((fp[8] - fm[8]) / (2 * d)))
    assert count == 2796
    assert worst[8] > 1.0 > worst[10] > 0.3              # measured 1.19 and 0.51
    assert worst[6] > 3.0 and worst[20] < 0.05           # measured 3.43 and 0.033
    assert 10.0 < worst[10] / worst[20] < 20.0           # ~2**4: fourth-order truncation error
    assert 1.0e4 < lam_max < 1.5e4 and lam_max * P.DT < 2.785 / 2      # measured 12.5e3 per day: lambda*dt = 1.05


# ----------------------------------------------------------------------------------------------- scheme 1 (round 5)
# cfg.scheme = 1: every reaction interval by Butcher's fifth-order scheme with 1, 2 or 4 steps chosen from the plant's own state
# (oracle/sbr_ref.py b5a_*, oracle/sbr_oracle.c b5a_interval) instead of ten RK4 substeps.  The same bars as
# scheme 0, on the same reference fixtures.
def _scheme1_params():
    return O.default_params(scheme=1)


@pytest.mark.parametrize("name", ["const_2_5", "random_b", "zeros", "scn0_c25", "scn4_phys", "scn5_c25", "scn7_phys", "ho_walk_s5", "ho_sine_s7"])
def test_scheme1_layer2_c_is_bit_identical_to_layer1(name, tables):
    means, stds = tables
    e = golden("sbros_" + name)
    b = O.OracleBatch(1, _scheme1_params())
    b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
    py = R.SbrOsRef(tables, integrator="rk4", scheme=1)
    py.reset(rnd=e["rnd"], scenario=_scn(e))
    n = int(e["n_calls"])
    for k in range(n):
        o, s, r, d = b.step(e["actions"][k][None])
        po, ps, pr, pd, _ = py.step(e["actions"][k])
        assert np.array_equal(o[0], np.r_[po[0], po[1]]) and np.array_equal(s[0], ps)
        assert abs(r[0] - pr) <= 4e-18 and bool(d[0]) == pd
        assert py.step_counts[-1] == b.envs["scheme_steps"][0]        # the same plan ...
        if k < n - 1:
            assert np.array_equal(b.envs["x"][0], py.x)                # ... and the same bits
    assert np.array_equal(b.envs["x"][0], py.x_after_idle) and b.envs["qw"][0] == py.qw
    assert set(py.step_counts) <= {1, 2, 4, 5}                        # 5: the knee of `zeros` (lam(0) span / 2.5 just above 4)


def test_scheme1_open_loop_every_interval_inside_gate():
    """One interval from the reference's own state, all 24 reference-captured episodes (9 661 intervals on which parity is
    defined): worst 0.225 of the gate (RK4 x 10: 0.509), with 10.2 right-hand-side evaluations per interval instead of 40."""
    p = _scheme1_params()
    worst, evals, count, hist = 0.0, 0, 0, {}
    for name in ALL_EPISODES:
        e = golden("sbros_" + name)
        nv = valid_calls(e)
        for i in range(len(e["iv_kind"])):
            if e["iv_call"][i] > nv:
                continue
            span = float(e["iv_t_end"][i] - e["iv_t_start"][i])
            x1, n = O.reaction_interval(e["iv_x_start"][i], span, float(e["iv_Kla"][i]), float(e["iv_EC"][i]), params=p)
            worst = max(worst, gate(x1, e["iv_x_end"][i]).max())
            hist[n] = hist.get(n, 0) + 1
            evals += 6 * n
            count += 1
    assert count == 9661 and set(hist) <= {1, 2, 4, 5} and hist.get(5, 0) < 300   # 5: only where lam(0) span / 2.5 >= 4 (`zeros`)
    assert worst <= 0.3, worst                                          # measured 0.2251 (So, aeration switch-on, scn3_c25)
    assert evals / count < 11.0, evals / count                          # measured 10.2 (slaved intervals take two steps)


@pytest.mark.parametrize("name", ALL_EPISODES)
def test_scheme1_closed_loop_inside_gate_of_the_reference_at_tight_tolerance(name, tables):
    """The closed-loop bar of scheme 0, unchanged: the whole chained episode against the unmodified reference at odeint
    rtol = atol = 1e-12, all 24 episodes (measured worst 0.36, Ss in `zeros`; 0.51 with RK4 x 10)."""
    means, stds = tables
    e = golden("sbros_%s_tight" % name)
    b = O.OracleBatch(1, _scheme1_params())
    b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
    xs, steps = [], []
    n, nv = int(e["n_calls"]), valid_calls(e)
    for k in range(n):
        b.step(e["actions"][k][None])
        xs.append(b.envs["x"][0].copy()); steps.append(int(b.envs["scheme_steps"][0]))
    xs = np.array(xs)
    m = min(nv, n - 1)
    assert gate(xs[:m], e["step_x_end"][:m]).max() <= 0.5, gate(xs[:m], e["step_x_end"][:m]).max()
    assert np.array_equal(b.envs["t"], e["step_t"][-1:])
    if nv == n:
        assert gate(xs[n - 1], e["term_x_after_idle"]).max() <= 1.0
        assert abs(b.envs["ret"][0] / float(e["episode_return"]) - 1) < 1e-5
        assert abs(b.envs["qw"][0] / float(e["term_Qw"]) - 1) < 1e-5
        assert set(steps) <= {1, 2, 4, 5}                               # the reference plant never needs more than five


def test_scheme1_takes_more_steps_where_four_would_be_unstable():
    """The stability rule of the knee: n = max(4, floor(lam(0) span / 2.5) + 1), so that the worst-case oxygen rate times the
    step stays below 2.5 (Butcher-5 is stable on the real axis up to 3.39).  The reference plant never needs more than four
    (lam(0) span / 4 <= 2.62 on every captured state); a plant with three times the biomass does, and stays accurate."""
    e = golden("sbros_const_2_5")
    i = int(np.where(e["iv_kind"] == 1)[0][0])                         # aeration switch-on: So = 0, Kla > 0
    x0 = e["iv_x_start"][i].copy()
    span, kla = float(e["iv_t_end"][i] - e["iv_t_start"][i]), float(e["iv_Kla"][i])
    x1, n = O.reaction_interval(x0, span, kla, 0.0)
    assert n == 4
    x0[5] *= 3.0; x0[6] *= 3.0
    x2, n2 = O.reaction_interval(x0, span, kla, 0.0)
    exact = O.rk4(0, x0, span, 320, kla, 0.0)
    assert 5 <= n2 <= 14 and gate(x2, exact).max() < 0.5, (n2, gate(x2, exact).max())
    assert O.reaction_interval(x0, span, kla, 0.0, scheme=0)[1] == -1
    x0[5] *= 1e6                                                       # absurd: the count is capped, the call returns
    assert O.reaction_interval(x0, span, kla, 0.0)[1] == 64
    # round 6: ... and a state OUTSIDE the model's domain (the Monod factor of Ss or Snh outside [0, 1], or NaN) does not raise the
    # count at all: the premise of the stability rule - an oxygen rate bounded by those factors - is gone, the state is garbage, and
    # one such env at 64 steps would make a whole batch wait for it (a launch lasts as long as its slowest wavefront)
    for idx, val in ((2, -10.5), (2, -1e3), (10, -1.2), (10, float("nan")), (2, float("nan"))):
        xg = e["iv_x_start"][i].copy()
        xg[5] *= 3.0; xg[6] *= 3.0
        xg[idx] = val
        assert O.reaction_interval(xg, span, kla, 0.0)[1] == 4, (idx, val)
    xg = e["iv_x_start"][i].copy(); xg[5] *= 3.0; xg[6] *= 3.0; xg[10] = -1e-300       # a rounding-level negative counts as inside
    assert 5 <= O.reaction_interval(xg, span, kla, 0.0)[1] <= n2


def test_scaled_mass_rk4_equals_concentration_form_rk4_far_below_the_truncation_error():
    """since round 4 both RK4 layers integrate dosing intervals in scaled-mass variables (rk4_reaction_w), so the
    device-vs-oracle comparison no longer checks that substitution by itself.  Pinned here: on every golden interval that doses
    carbon, RK4 x 10 on w = c V/V0 and RK4 x 10 on c (the reference's own variables) agree to < 1e-6 of the parity gate (measured 2.8e-8), six
    orders of magnitude below RK4's own truncation error (0.044) on the same interval (RK4 x 10 against RK4 x 20)."""
    worst, worst_trunc, n = 0.0, 0.0, 0
    for name in ("const_2_5", "random_a", "max", "scn4_phys", "scn5_c1_7"):
        e = golden("sbros_" + name)
        for i in np.where(e["iv_EC"] != 0)[0][::3]:
            x0, kla, ec = e["iv_x_start"][i], float(e["iv_Kla"][i]), float(e["iv_EC"][i])
            t0, t1 = float(e["iv_t_start"][i]), float(e["iv_t_end"][i])
            xc = R.rk4(R.rhs_reaction, x0, t0, t1, 10, (kla, ec))
            xw = R.rk4_reaction_w(x0, t0, t1, 10, kla, ec)
            x20 = R.rk4(R.rhs_reaction, x0, t0, t1, 20, (kla, ec))
            worst = max(worst, gate(xw, xc).max())
            worst_trunc = max(worst_trunc, gate(xc, x20).max())
            n += 1
    assert n > 100 and worst < 1e-6 and worst_trunc > 1e4 * worst, (n, worst, worst_trunc)      # measured 2.8e-8 against 0.044


def test_scheme1_on_the_intervals_of_the_per_cycle_env_fresh_and_carried_over(tables):
    """A second population for the adaptive scheme's plan (round 5): every reaction / idle interval of SBR-v2 cycles with random
    set-points, fresh AND carried over (concentrated sludge after the draw, oxygen left over from the aerated idle phase, ammonia
    exhausted by an 8 g/m3 set-point) - states no SBROS-v1 fixture holds.  Each interval on its own, scheme 1 against RK4 x 160:
    inside 0.3 of the gate (measured 0.13 over 36 000 intervals of 24 envs x 3 cycles; here 4 envs x 2 cycles).
    The FILL intervals of those cycles are the counter-example that keeps the fill phase with RK4 under either scheme: planned
    on their own they were up to 70 gates off (the inflow raises Ss and Snh severalfold within an interval)."""
    import importlib.util
    from conftest import ROOT
    import os
    spec = importlib.util.spec_from_file_location("cycle_intervals", os.path.join(ROOT, "scripts", "analysis", "cycle_intervals.py"))
    ci = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ci)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        pop = ci.collect(n_envs=4, seed=8, cycles=2)
    worst, count, hist = 0.0, 0, {}
    for j in np.where(pop["kind"] == 2)[0]:
        x0, span, kla = pop["X"][:, j].copy(), float(pop["span"][j]), float(pop["kla"][j])
        x1, n = O.reaction_interval(x0, span, kla, 0.0, scheme=1)
        worst = max(worst, gate(x1, O.rk4(0, x0, span, 160, kla, 0.0)).max())
        hist[n] = hist.get(n, 0) + 1
        count += 1
    assert count > 3500 and worst < 0.3, (count, worst)
    assert set(hist) <= {1, 2, 4, 5, 6} and hist.get(4, 0) > 50    # knees are in the population


def test_scheme1_plan_on_states_far_from_the_reference_regime():
    """A robustness probe of the plan, not a parity statement: 3 000 random plant states far beyond anything the fixtures hold
    (sludge 0.4 - 3 x the reference's, substrate from exhausted to shock-loaded, oxygen from 0 to saturation, every Kla and EC;
    scripts/analysis/plan_probe.py), one interval each against RK4 x 320.  Such states sit on several Monod knees at once and ten
    RK4 substeps themselves miss the gate on ~2.4 % of them; scheme 1 must be no worse than that by more than a third, and must
    not blow up where RK4 x 10 is fine (round 5: the first version of the plan took the oxygen rate at the START levels of Ss and
    Snh and was unstable when carbon dosing raised Ss within the interval - 4 000 gates; the plan now projects them)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("plan_probe", os.path.join(ROOT, "scripts", "analysis", "plan_probe.py"))
    pp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pp)
    rs = np.random.RandomState(1)
    span = (0.25 + P.T_DELTA) - 0.25
    p1 = _scheme1_params()
    g1, g0 = [], []
    for _ in range(3000):
        x, kla, ec = pp.sample(rs)
        ex = O.rk4(0, x, span, 320, kla, ec)
        if not (np.isfinite(ex).all() and (ex[[2, 4, 5, 8, 9, 10]] > -1e-9).all()):
            continue                                        # the fine solution itself leaves the model's domain
        g1.append(gate(O.reaction_interval(x, span, kla, ec, params=p1, scheme=1)[0], ex).max())
        g0.append(gate(O.rk4(0, x, span, 10, kla, ec), ex).max())
    g1, g0 = np.array(g1), np.array(g0)
    assert len(g1) > 2800
    assert (g1 > 1).sum() <= 1.35 * (g0 > 1).sum() + 5, ((g1 > 1).sum(), (g0 > 1).sum())          # measured 91 against 68
    assert np.percentile(g1, 99) < 8 and g1.max() < 200                                             # measured 4.3 and 39 (RK4 x 10: 5.8, 3 000)
    assert (g1[g0 < 0.1] > 30).sum() == 0                   # no blow-up where ten RK4 substeps are accurate (measured: worst 6.9)


# ---------------------------------------------------------------------------------------------------- held-out episodes (round 6)
@pytest.mark.parametrize("scheme", [1, 0])
def test_heldout_episodes_open_and_closed_loop(scheme, tables):
    """VERDICT r5 item 3(a).  The plan thresholds of cfg.scheme = 1 (0.3 / 1.0 / 2.5, the slaved test, two slaved steps) were fitted on
    the 24 episodes of ALL_EPISODES; these ten reference episodes were captured AFTERWARDS and are excluded from any fitting
    (oracle/gen_golden.py heldout_cases: new influent seeds, the reference's own random-walk action model through its
    get_available_actions, set-points held 20 calls, a sinusoidal DO set-point sweeping the oxygen knee).  The bars are the fitted
    set's: open loop (one interval from the reference's own state, against its LSODA end state) and closed loop (the whole chained
    episode against the reference at odeint rtol = atol = 1e-12) within 0.6 of the gate.  Measured: scheme 1 open loop 0.222,
    closed loop 0.373 (Ss of `ho_walk_s5`, call 52 - the episode on which the reference's OWN default-tolerance run is 13 gates
    from its own 1e-12 run, see the next test); RK4 x 10: 0.470 and 0.254.  A value above 0.6 here is a finding to be
    reported, not a reason to retune the thresholds on these episodes."""
    means, stds = tables
    p = O.default_params(scheme=scheme)
    worst_open, worst_closed, worst_term = {}, {}, {}
    for name in HELDOUT_EPISODES:
        e, t = golden("sbros_" + name), golden("sbros_%s_tight" % name)
        nv = min(valid_calls(e), valid_calls(t))
        wo = 0.0
        for i in range(len(e["iv_kind"])):
            if e["iv_call"][i] > nv:
                continue
            span = float(e["iv_t_end"][i] - e["iv_t_start"][i])
            x1, n = O.reaction_interval(e["iv_x_start"][i], span, float(e["iv_Kla"][i]), float(e["iv_EC"][i]), params=p, scheme=scheme)
            wo = max(wo, gate(x1, e["iv_x_end"][i]).max())
            assert n == -1 if scheme == 0 else 1 <= n <= 5
        worst_open[name] = wo
        b = O.OracleBatch(1, O.default_params(scheme=scheme))
        b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
        assert gate(b.envs["x"][0], e["x_postfill"]).max() <= 0.01        # the fill phase (RK4 x 252 under either scheme): measured 0.003
        wc = 0.0
        for k in range(min(nv, 462)):
            b.step(t["actions"][k][None])
            wc = max(wc, gate(b.envs["x"][0], t["step_x_end"][k]).max())
        worst_closed[name] = wc
        if nv == 463:
            b.step(t["actions"][462][None])
            worst_term[name] = gate(b.envs["x"][0], t["term_x_after_idle"]).max()
            assert abs(b.envs["ret"][0] / float(t["episode_return"]) - 1) < 1e-5 and abs(b.envs["qw"][0] / float(t["term_Qw"]) - 1) < 1e-5
        else:
            assert name == "ho_held20_s1" and nv == 295                   # the low-ammonia scenario leaves the domain of parity, as scn1_* do
    print("[info] held-out episodes, cfg.scheme = %d: open loop worst %.3f (%s), closed loop vs the reference at 1e-12 worst %.3f (%s), "
          "terminal state worst %.3f" % (scheme, max(worst_open.values()), max(worst_open, key=worst_open.get), max(worst_closed.values()),
                                         max(worst_closed, key=worst_closed.get), max(worst_term.values())))
    assert max(worst_open.values()) <= 0.6 and max(worst_closed.values()) <= 0.6 and max(worst_term.values()) <= 0.6


def test_heldout_walk_episodes_and_the_reference_own_integrator_noise(tables):
    """What the held-out random-walk episodes add to DESIGN.md 4.3's finding about `random_b` and `zeros`: under the reference's own
    action model the NO3 set-point sits for long stretches where the dosing PID is unsaturated, and the closed loop amplifies
    LSODA's default-tolerance error (1.5e-8 per interval) - the reference's default run of `ho_walk_s5` ends up 13 gates (Ss) from
    its OWN run at 1e-12, `ho_walk_s6` / `ho_walk_s7` 1.3 gates.  No integrator other than the bit-identical LSODA run (layer 1)
    can follow such a default-tolerance trajectory; the C oracle (either scheme) is as far from it as the reference's tight run
    is, to 5 %, and inside the gate of the tight run (previous test)."""
    means, stds = tables
    for name, lo, hi in (("ho_walk_s5", 8.0, 20.0), ("ho_walk_s6", 1.0, 2.0), ("ho_walk_s7", 1.0, 2.0), ("ho_walk_s4", 0.0, 0.3)):
        e, t = golden("sbros_" + name), golden("sbros_%s_tight" % name)
        own = gate(t["step_x_end"][:462], e["step_x_end"][:462])
        assert lo <= own.max() <= hi, (name, own.max())
        for scheme in (1, 0):
            b = O.OracleBatch(1, O.default_params(scheme=scheme))
            b.reset(b.mix(means, stds, [_scn(e)], e["rnd"][None]))
            xs = []
            for k in range(462):
                b.step(e["actions"][k][None])
                xs.append(b.envs["x"][0].copy())
            d = gate(np.array(xs), e["step_x_end"][:462])
            assert d.max() <= 1.05 * own.max() + 0.4, (name, scheme, d.max(), own.max())
            assert np.delete(d, 2, axis=1).max() <= max(1.0, 1.05 * np.delete(own, 2, axis=1).max() + 0.05)      # it is Ss (and what Ss drives)


def test_scheme1_under_perturbed_kinetic_constants():
    """the plan of cfg.scheme = 1 was validated with the reference's kinetic constants; what does it do for a
    plant configured with FASTER kinetics (muH, muA, kh each up to 4 x), where every mode of the system is stiffer?  3 000 random
    plant states with reference-like sludge (scripts/analysis/plan_probe.py), one interval each against RK4 x 1280, both schemes.
    Measured: scheme 1 misses the gate on 134 states (11 beyond 30 gates), ten RK4 substeps on 420 (239 beyond 30 - their step
    is fixed at h = dt, which is unstable once the oxygen rate times dt exceeds 2.785, while scheme 1's count grows with it).
    Scheme 1 must not be worse than scheme 0 there.  The state-dependent stability floor for the Ss / Snh / Sno modes that
    round 6 built and did not adopt (oracle study knob; 0.2 us per k_step call) would halve what is left: asserted too, so
    that the number stays on record."""
    import ctypes as C
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("plan_probe", os.path.join(ROOT, "scripts", "analysis", "plan_probe.py"))
    pp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pp)
    span = (0.25 + P.T_DELTA) - 0.25
    knobs = O.lib().sbro_set_plan_knobs
    knobs.argtypes = [C.c_double]

    def run(guard):
        rs = np.random.RandomState(3)
        knobs(2.5 if guard else 0.0)
        g1, g0 = [], []
        try:
            for _ in range(1500):
                x, kla, ec = pp.sample(rs)
                x[5], x[6] = rs.uniform(1500, 3000), rs.uniform(80, 200)
                p1, p0 = O.default_params(scheme=1), O.default_params(scheme=0)
                f = 10 ** rs.uniform(0, np.log10(4.0), 3)
                for p in (p1, p0):
                    p.muH *= f[0]; p.muA *= f[1]; p.kh *= f[2]
                ex = O.rk4(0, x, span, 1280, kla, ec, params=p0)
                if not (np.isfinite(ex).all() and (ex[[2, 4, 5, 8, 9, 10]] > -1e-9).all()):
                    continue
                x1 = O.reaction_interval(x, span, kla, ec, params=p1, scheme=1)[0]
                g1.append(gate(x1, ex).max() if np.isfinite(x1).all() else 1e30)
                r = O.rk4(0, x, span, 10, kla, ec, params=p0)
                g0.append(gate(r, ex).max() if np.isfinite(r).all() else 1e30)
        finally:
            knobs(0.0)
        return np.array(g1), np.array(g0)
    g1, g0 = run(False)
    assert len(g1) > 1350
    print("[info] kinetic constants up to 4 x: scheme 1 misses the gate on %d of %d states (%d beyond 30 gates, worst %.3g); RK4 x 10 on %d "
          "(%d beyond 30)" % ((g1 > 1).sum(), len(g1), (g1 > 30).sum(), g1.max(), (g0 > 1).sum(), (g0 > 30).sum()))
    assert (g1 > 1).sum() <= 0.6 * (g0 > 1).sum() and (g1 > 30).sum() <= 0.2 * (g0 > 30).sum() + 2
    g1g, _ = run(True)
    print("[info] ... with the stability floor of the other modes (not adopted): %d (%d beyond 30, worst %.3g)" % ((g1g > 1).sum(), (g1g > 30).sum(), g1g.max()))
    assert (g1g > 30).sum() <= (g1 > 30).sum() and g1g.max() <= g1.max()
