"""Parity tests proper: the HIP path (through the C ABI of include/sbr_amd.h) against the CPU oracle and
the golden fixtures.  Run on the GPU box with `pytest -m gpu`; nothing here reads /root/reference.

Tolerances are set from measurements on MI355X (profiles/r01_explore.txt), ~100x above the worst value
seen, and are written next to each assertion.  GPU vs oracle differences come only from FMA contraction
and summation order (both sides are fp64 RK4 with identical inputs: the oracle is always fed exactly the action
values the device sees - float64 where the test compares with the golden vectors, float32-rounded elsewhere).
"""
import os

import numpy as np
import pytest
from conftest import BENCH_SCENARIOS, EPISODES, HELDOUT_EPISODES, SCENARIO_EPISODES, gate, golden, obs_tolerance, valid_calls

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import sbr_oracle as O  # noqa: E402  (the checker, never the thing under test)

CLOSED_LOOP_OK = ["const_2_5", "random_a", "max", "det_influent"]


@pytest.fixture(scope="module")
def G():
    import gym_sbr2_amd
    from gym_sbr2_amd import _capi
    assert torch.cuda.is_available(), "these tests need the GPU box"
    lib = _capi.load()                            # builds it with hipcc if the snapshot lacks it; never a fallback
    assert _capi.library_path().endswith(os.path.join("gym_sbr2_amd", "lib", "libsbr_amd.so"))   # the in-tree .so is what runs
    assert any("libsbr_amd.so" in l for l in open("/proc/self/maps"))
    assert lib.sbr_device_count() >= 1
    return gym_sbr2_amd


def _np(t):
    return t.detach().cpu().numpy()


def _plans_agree(ctrl, ora, where=None, pick=None):
    """the plan of cfg.scheme = 1 - Butcher-5 step count and the slaved bit of the env's last interval - as the
    DEVICE reports it (SBR_C_PLAN, from the meta row) equals the oracle's for the same call.  A decision that flipped (a rounded
    double on the other side of 0.3 / 1.0 / 1e-9) would show as a ~1e-2-gate mismatch otherwise indistinguishable from a wrong kernel."""
    from gym_sbr2_amd import _capi
    dev = np.asarray(ctrl[_capi.C_PLAN]).astype(np.int64)
    if pick is not None:
        dev = dev[pick]
    ref = ora.envs["scheme_plan"].astype(np.int64) & 0xff
    if where is not None:
        dev, ref = dev[where], ref[where]
    bad = np.nonzero(dev != ref)[0]
    assert bad.size == 0, ("plan flipped on %d envs; first: device %d, oracle %d" % (bad.size, dev[bad[0]], ref[bad[0]]))
    return dev


def test_native_library_is_loaded_and_fails_loudly_on_bad_config(G):
    import ctypes as C
    from gym_sbr2_amd import _capi
    lib = _capi.load()
    assert b"gfx950" in lib.sbr_version()
    cfg = _capi.default_config()
    cfg.t_delta = cfg.dt * 7
    h = C.c_void_p()
    assert lib.sbr_create(4, 0, 0, C.byref(cfg), C.byref(h)) == -1 and b"t_delta" in lib.sbr_last_error(None)
    assert lib.sbr_create(0, 0, 0, None, C.byref(h)) == -1
    assert lib.sbr_create(4, 99, 0, None, C.byref(h)) == -1
    # the guard of the dosing integrator's 1/s series (EC_max * t_delta <= 1e-4 of the reactor volume) and the scheme
    cfg = _capi.default_config(); cfg.EC_max = 2.0
    assert lib.sbr_create(4, 0, 0, C.byref(cfg), C.byref(h)) == -1 and b"EC_max" in lib.sbr_last_error(None)
    cfg = _capi.default_config(); cfg.EC_max = 1e-4 * cfg.IV / cfg.t_delta * 0.99
    assert lib.sbr_create(4, 0, 0, C.byref(cfg), C.byref(h)) == 0 and lib.sbr_destroy(h) == 0
    cfg = _capi.default_config(); cfg.scheme = 2
    assert lib.sbr_create(4, 0, 0, C.byref(cfg), C.byref(h)) == -1 and b"scheme" in lib.sbr_last_error(None)
    cfg = _capi.default_config(); cfg.reserved_ = 7          # 'must be 0' is enforced
    assert lib.sbr_create(4, 0, 0, C.byref(cfg), C.byref(h)) == -1 and b"reserved_" in lib.sbr_last_error(None)
    env = G.SbrOSVec(4)
    # sbr_query: the library's own launch decisions for this handle and device (bench.py used to repeat the thresholds)
    one_wave = env.query(_capi.Q_ONE_WAVE_ENVS)
    assert one_wave == torch.cuda.get_device_properties(0).multi_processor_count * 256
    assert (env.query(_capi.Q_STEP_BLOCK), env.query(_capi.Q_STEP_WAVES), env.query(_capi.Q_ROLLOUT_WAVES), env.query(_capi.Q_SCHEME)) == (64, 1, 1, 1)
    assert env.query(_capi.Q_STEP_SMALL_BATCH_ENVS) == 49152 and env.query(_capi.Q_STEP_TWO_WAVES_ABOVE_ENVS) == max(one_wave, 49152)
    assert env.query(_capi.Q_FUSED_ONE_WAVE_MAX_ENVS) == one_wave + one_wave // 2 and env.query(_capi.Q_RESET_BLOCK) == 256
    out_q = C.c_int64()
    assert lib.sbr_query(env._h, 99, C.byref(out_q)) == -1 and lib.sbr_query(None, 0, C.byref(out_q)) == -1
    big = G.SbrOSVec(one_wave + 256)
    assert (big.query(_capi.Q_STEP_BLOCK), big.query(_capi.Q_STEP_WAVES), big.query(_capi.Q_RESET_BLOCK)) == (256, 2, 512)
    big.close()
    with pytest.raises(ValueError):
        env.step(torch.zeros(3, 2))
    # the call counter shares a row with the plan code, the flags and the status bits (an integer below 2^31 held in a double): it
    # saturates at 2^17 - 1 calls per episode (until round 6: 2^25 - 1; an episode of the reference has 463), an imported value
    # beyond that is clamped, the plan code beside it is kept (0 .. 255), and stepping on neither wraps nor disturbs the flags
    env.reset(seed=1)
    x_s, c_s = env.get_state()
    c_s[_capi.C_STEPS] = torch.tensor([5.0, 131070.0, 131071.0, 1e9], dtype=torch.float64, device="cuda")
    c_s[_capi.C_PLAN] = torch.tensor([0.0, 4.0, 130.0, 999.0], dtype=torch.float64, device="cuda")
    env.set_state(x_s, c_s)
    _, c_r = env.get_state()
    assert _np(c_r[_capi.C_STEPS]).tolist() == [5, 131070, 131071, 131071] and _np(c_r[_capi.C_PLAN]).tolist() == [0, 4, 130, 255]
    env.step(torch.full((4, 2), 2.0, device="cuda")); env.step(torch.full((4, 2), 2.0, device="cuda"))
    _, c_r = env.get_state()
    assert _np(c_r[_capi.C_STEPS]).tolist() == [7, 131071, 131071, 131071] and _np(c_r[_capi.C_DONE]).tolist() == [0, 0, 0, 0]
    assert _np(c_r[_capi.C_STATUS]).tolist() == [0, 0, 0, 0] and np.all((_np(c_r[_capi.C_PLAN]).astype(int) & 127) >= 1)
    # an env that was never reset is inert: done = 1, reward 0
    env.close(); env = G.SbrOSVec(4)
    _, _, r, d = env.step(torch.zeros(4, 2))
    assert _np(d).tolist() == [1, 1, 1, 1] and _np(r).tolist() == [0, 0, 0, 0]
    # nulls at the ABI: every output of sbr_step is optional, the action is not; bad rows are refused with a message
    env.reset(seed=1)
    a = torch.full((4, 2), 2.0, device="cuda")
    assert lib.sbr_step(env._h, a.data_ptr(), None, None, None, None, None) == 0
    torch.cuda.synchronize()
    assert _np(env.ctrl_row(_capi.C_STEPS)).tolist() == [1, 1, 1, 1]
    assert lib.sbr_step(env._h, None, None, None, None, None, None) == -1 and b"action" in lib.sbr_last_error(env._h)
    out = torch.empty(4, dtype=torch.float64, device="cuda")
    assert lib.sbr_get_ctrl_row(env._h, _capi.NCTRL, out.data_ptr(), None) == -1
    assert lib.sbr_get_ctrl_row(env._h, -1, out.data_ptr(), None) == -1
    assert lib.sbr_rollout(env._h, -3, 0, None, None, None) == -1
    # sbr_eval_substeps: kinds 0..3 only, a control interval needs ec, the fill phase its loading vector, n_sub >= 1
    x0, v = torch.ones(4, 14, dtype=torch.float64, device="cuda"), torch.ones(4, dtype=torch.float64, device="cuda")
    xs = torch.empty(4, 11, 14, dtype=torch.float64, device="cuda"); dxs = torch.empty_like(xs)
    P = lambda t: t.data_ptr()                                    # noqa: E731
    assert lib.sbr_eval_substeps(env._h, 0, 4, 10, P(x0), P(v), P(v), None, P(v), P(xs), P(dxs), None) == 0
    assert lib.sbr_eval_substeps(env._h, 4, 4, 10, P(x0), P(v), P(v), None, P(v), P(xs), P(dxs), None) == -1
    assert lib.sbr_eval_substeps(env._h, 0, 4, 10, P(x0), P(v), None, None, P(v), P(xs), P(dxs), None) == -1
    assert lib.sbr_eval_substeps(env._h, 1, 4, 10, P(x0), P(v), None, None, P(v), P(xs), P(dxs), None) == -1
    assert lib.sbr_eval_substeps(env._h, 2, 4, 0, P(x0), P(v), None, None, P(v), P(xs), P(dxs), None) == -1
    assert b"sbr_eval_substeps" in lib.sbr_last_error(env._h)
    torch.cuda.synchronize()
    # scenario ids outside 0..7 are clamped, never used as an index
    z = np.zeros((4, 48))
    env.reset(scenario=np.array([-5, 0, 7, 100], dtype=np.int32), rnd=z)
    got = _np(env.influent())
    env.reset(scenario=np.array([0, 0, 7, 7], dtype=np.int32), rnd=z)
    assert np.array_equal(got, _np(env.influent()))
    env.close()


def test_rhs_known_answers_on_device(G):
    k = golden("rhs_kat")
    env = G.SbrOSVec(len(k["X"]), out_dtype=torch.float64)
    for kind, key in [(0, "d_reaction"), (1, "d_filling"), (2, "d_idle")]:
        ec = k["ec"] if kind == 0 else np.zeros_like(k["ec"])
        d = _np(env.eval_rhs(kind, k["X"], k["kla"], ec, k["loading"] if kind == 1 else None))
        rel = np.abs(d - k[key]) / np.abs(k[key]).max(axis=1, keepdims=True)
        assert rel.max() < 5e-14, (key, rel.max())           # measured 4.0e-16
    env.close()


def test_influent_mix_and_device_normals(G, tables):
    means, stds = tables
    ik = golden("influent_kat")
    n = len(ik["scenario"])
    env = G.SbrOSVec(n, out_dtype=torch.float64)
    env.reset(scenario=ik["scenario"], rnd=ik["rnd"])
    got = _np(env.influent()).T
    assert np.abs(got[:, 1:] - ik["mixed"][:, 1:]).max() < 1e-11     # measured 2.8e-14 (values up to 260)
    cfg = env.cfg
    assert np.all(got[:, 0] == (cfg.WV - cfg.IV) / cfg.T_fill)       # entry 0 = Qin / T_fill (:287)
    # Philox + Box-Muller on the device = the oracle's restatement, keyed by GLOBAL env id
    z = _np(env.draw_normals(7))
    assert np.abs(z - O.OracleBatch(n).normals(7)).max() < 1e-13     # measured 4.4e-16
    env2 = G.SbrOSVec(4, first_env_id=9)
    assert np.array_equal(_np(env2.draw_normals(7)), z[9:13])
    # reset without rnd uses exactly those normals
    env.reset(seed=7, scenario=ik["scenario"])
    ora = O.OracleBatch(n)
    assert np.abs(_np(env.influent()).T[:, 1:] - ora.mix(means, stds, ik["scenario"], z)[:, 1:]).max() < 1e-11
    env.close(); env2.close()


def _run_golden_batch(G, tables, out_dtype, scheme=1):
    from gym_sbr2_amd import _capi
    means, stds = tables
    E = [golden("sbros_" + n) for n in EPISODES]
    n, ncall = len(E), int(E[0]["n_calls"])
    rnd = np.stack([e["rnd"] for e in E])
    # float64 actions, what the reference's step() receives: with Kc_EC = 100 against an EC range of 5e-4 a
    # float32-rounded set-point (15 +- 9e-7) moves an unsaturated EC by up to 9e-5 (float32 actions are covered
    # by the 4096-env and rollout tests, against the oracle fed the same rounded values)
    acts = np.stack([e["actions"][:ncall] for e in E], axis=1).astype(np.float64)
    cfg = _capi.default_config(); cfg.scheme = scheme
    env = G.SbrOSVec(n, out_dtype=out_dtype, action_dtype=torch.float64, config=cfg)
    ora = O.OracleBatch(n, O.default_params(scheme=scheme))
    obs0 = _np(env.reset(rnd=rnd)).copy()
    oobs0 = ora.reset(ora.mix(means, stds, [6] * n, rnd))
    return E, env, ora, acts, obs0, oobs0, ncall


@pytest.mark.parametrize("scheme", [1, 0])
def test_six_golden_episodes_against_oracle_and_reference(G, tables, scheme):
    """Both integrators: cfg.scheme = 1 (the default) and 0 (ten RK4 substeps, what rounds 1-4 shipped: its tighter agreement with the
    reference's rewards stays pinned - ADVICE r5)."""
    from gym_sbr2_amd import _capi
    E, env, ora, acts, obs0, oobs0, ncall = _run_golden_batch(G, tables, torch.float64, scheme)
    n = len(E)
    x, ctrl = env.get_state()
    assert gate(_np(x).T, ora.envs["x"]).max() < 1e-6                    # post-fill, measured 9.7e-10
    # vs the reference: the fill phase is RK4 x 252 under either scheme, measured 1.6e-3 of the gate.  (Rounds 5's bounds here - 1.0 of
    # the gate, the reset observation within the gate-derived ~2e-5 - were set while an adaptive fill phase was being tried and
    # outlived it: VERDICT r5 weak 2.  A fill-phase regression of 6 x now fails.)
    assert gate(_np(x).T, np.stack([e["x_postfill"] for e in E])).max() < 0.01
    assert np.abs(obs0 - oobs0).max() < 1e-11                            # measured 4.4e-15
    for i, e in enumerate(E):
        assert np.abs(obs0[i] - np.r_[e["reset_obs_DO"], e["reset_obs_EC"]]).max() < 1e-6      # measured 2e-8
    T = [golden("sbros_%s_tight" % name) for name in EPISODES]      # the reference itself at odeint rtol = atol = 1e-12
    worst_gold, worst_tight = np.zeros(n), np.zeros(n)
    ec_prev_dev = 0.0
    for c in range(ncall):
        o, s, r, d = env.step(torch.from_numpy(acts[c]).cuda())
        oo, os_, orr, od = ora.step(acts[c])
        x, ctrl = env.get_state()
        x, ctrl = _np(x).T, _np(ctrl)
        assert np.array_equal(_np(d), od) and np.array_equal(_np(d), [int(e["step_done"][c]) for e in E])
        assert gate(x, ora.envs["x"]).max() < 1e-6, c                    # measured 5.6e-9  (= 6e-14 relative)
        assert np.abs(_np(o) - oo).max() < 1e-10 and np.abs(_np(s) - os_).max() < 1e-10     # measured 3.4e-13
        assert np.abs(_np(r) - orr).max() < 1e-12                        # measured 1.2e-15
        assert np.abs(ctrl[_capi.C_KLA_LAST] - ora.envs["kla_last"]).max() < 1e-9          # measured 3.0e-12
        assert np.abs(ctrl[_capi.C_EC_LAST] - ora.envs["ec_last"]).max() < 1e-14           # measured 7e-17
        assert np.abs(ctrl[_capi.C_IE_DO] - ora.envs["ie_do"]).max() < 1e-14
        assert np.abs(ctrl[_capi.C_IE_EC] - ora.envs["ie_ec"]).max() < 1e-14
        assert np.array_equal(ctrl[_capi.C_T], ora.envs["t"])            # the time recurrence is exact
        assert np.abs(ctrl[_capi.C_KLA_HIST0:_capi.C_KLA_HIST0 + 10].T - ora.envs["kla_hist"]).max() < 1e-9
        plans = _plans_agree(ctrl, ora)                                  # free-running on both sides, states equal to 1e-8 of the gate
        assert (np.all((plans & 127) >= 1) and np.all((plans & 127) <= 8)) if scheme == 1 else np.all(plans == 0)
        if c < ncall - 1:
            worst_gold = np.maximum(worst_gold, [gate(x[i], E[i]["step_x_end"][c]).max() for i in range(n)])
            worst_tight = np.maximum(worst_tight, [gate(x[i], T[i]["step_x_end"][c]).max() for i in range(n)])
            # rewards, all six episodes.  `zeros` (index 3) gets 5e-6: its NO3-PID output is unsaturated for long stretches, where
            # EC carries Kc = 100 times the deviation of Sno (a third of a gate = 7e-5 moves EC by 14 % of its range) into the
            # cost term of the reward (measured: 2.0e-6 under scheme 1, 3e-7 under scheme 0; every other episode < 4e-10)
            dr = np.abs(_np(r) - [t["step_reward"][c] for t in T])
            assert np.delete(dr, 3).max() < 5e-7 and dr[3] < (5e-6 if scheme == 1 else 5e-7)       # scheme 0: measured 3e-7
            # ... and the 5e-6 of `zeros` is what its dosing deviation implies, call by call: the reward is
            # (1 - EQI2^2 - OCI^2)/473 with OCI = AE_OCI + EC_OCI <= 3.5 and EC_OCI = EC_conc sum(EC[-rows:-1]) td / (span 1000) =
            # 4320 x (a weighted mean of this call's and the previous call's EC) for rows = 10 (module_reward_EQIOCI.py:70-107), so
            # |d reward| <= 2 x 3.5 x 4320 / 473 x |d EC| + the state part (< 5e-7) = 64 |d EC| + 5e-7; the velocity-form NO3-PID
            # (EC += Kc_EC e + ..., Kc_EC = 100, gym_SBR_oneshot.py:2006-2025) integrates 100 x the deviation of Sno into EC while
            # it is unsaturated: |d EC| reaches 3e-8 in `zeros`, i.e. 2e-6 in the reward
            d_ec = max(abs(ctrl[_capi.C_EC_LAST][3] - T[3]["step_EC"][c]), abs(ec_prev_dev))
            assert dr[3] <= 64.0 * d_ec + 5e-7, (c, dr[3], d_ec)
            ec_prev_dev = ctrl[_capi.C_EC_LAST][3] - T[3]["step_EC"][c]
            for i, e in enumerate(E):      # rewards / observations against the reference itself
                if EPISODES[i] in CLOSED_LOOP_OK:
                    # reward = (1 - S)/473 with S = EQI2^2 + OCI^2 <= ~5: a 1e-5 state error gives <= 2e-5*S/473 ~ 2e-7
                    assert abs(_np(r)[i] - e["step_reward"][c]) < 5e-7
    assert np.array_equal(ctrl[_capi.C_STEPS], np.full(n, float(ncall))) and np.all(ctrl[_capi.C_DONE] == 1.0)
    assert np.all(ctrl[_capi.C_STATUS] == 0) and np.all(ora.envs["status"] == 0)   # the reference episodes stay physical
    for i, name in enumerate(EPISODES):
        e = E[i]
        if name in CLOSED_LOOP_OK:         # the north-star bar: trajectories within 1e-5 of the reference integrator
            assert worst_gold[i] <= 1.0, (name, worst_gold[i])
            assert gate(x[i], e["term_x_after_idle"]).max() <= 1.0
            assert abs(ctrl[_capi.C_RETURN][i] / float(e["episode_return"]) - 1) < 1e-5
            assert abs(ctrl[_capi.C_QW][i] / float(e["term_Qw"]) - 1) < 1e-5
        else:                              # the reference's own default-tolerance LSODA noise exceeds the gate here (Ss only)
            print("[info] %s: device vs reference at default odeint tolerance: %.3f of the gate" % (name, worst_gold[i]))
            # bounded all the same: no further than the reference's own tight-tolerance run is from it (2.58 / 1.35), + 5 %
            own = gate(T[i]["step_x_end"][:ncall - 1], e["step_x_end"][:ncall - 1]).max()
            assert worst_gold[i] < 3.0 and worst_gold[i] <= 1.05 * own + 0.05, (name, worst_gold[i], own)
        # the closed-loop bar on ALL six episodes: the reference itself at tight integrator tolerance
        t = T[i]
        assert worst_tight[i] <= 1.0, (name, worst_tight[i])                       # measured worst 0.51 (So, random_b)
        assert gate(x[i], t["term_x_after_idle"]).max() <= 1.0
        assert abs(ctrl[_capi.C_RETURN][i] / float(t["episode_return"]) - 1) < 1e-5
        assert abs(ctrl[_capi.C_QW][i] / float(t["term_Qw"]) - 1) < 1e-5
        assert abs(ctrl[_capi.C_QW][i] / ora.envs["qw"][i] - 1) < 1e-11           # measured 7e-14
    # a finished env ignores further calls until reset
    _, _, r, d = env.step(torch.from_numpy(acts[0]).cuda())
    assert np.all(_np(d) == 1) and np.all(_np(r) == 0)
    x2, _ = env.get_state()
    assert np.array_equal(_np(x2).T, x)
    env.close()


def test_float32_outputs_are_the_rounded_float64_outputs(G, tables):
    """The float32 kernel must differ from the float64 one ONLY in the final cast of obs/state/reward: the plant is
    float64 in both.  Different template instantiations may contract FMAs differently, so the plants are compared
    at gate 1e-6 (= 1e-11 relative) - float32 leaking into the plant would show as ~1e-2."""
    E, env64, _, acts, obs64, _, _ = _run_golden_batch(G, tables, torch.float64)
    _, env32, _, _, obs32, _, _ = _run_golden_batch(G, tables, torch.float32)
    assert obs32.dtype == np.float32 and np.allclose(obs32, obs64, rtol=1.2e-7, atol=1e-30)
    for c in range(70):
        a = torch.from_numpy(acts[c]).cuda()
        o64, s64, r64, d64 = env64.step(a)
        o32, s32, r32, d32 = env32.step(a)
        assert o32.dtype == torch.float32 and s32.dtype == torch.float32 and r32.dtype == torch.float32
        assert np.allclose(_np(o32), _np(o64), rtol=1.2e-7, atol=1e-30) and np.allclose(_np(s32), _np(s64), rtol=1.2e-7, atol=1e-30)
        assert np.allclose(_np(r32), _np(r64), rtol=1.2e-7, atol=1e-30) and np.array_equal(_np(d32), _np(d64))
    x64, c64 = env64.get_state()
    x32, c32 = env32.get_state()
    assert gate(_np(x32).T, _np(x64).T).max() < 1e-6
    assert np.array_equal(_np(c32)[0], _np(c64)[0])                  # time
    env64.close(); env32.close()


def test_open_loop_intervals_from_reference_states(G):
    """North star: 'match the reference odeint step on identical initial states'.  Every one of the 466
    intervals of an episode is started from the reference's own state and controller memory (set_state),
    one step() is run on the device and the end state must be inside the gate of the reference's.
    Round 5: on ALL EIGHT influent scenarios (the 18 scenario episodes captured from the reference with the harness
    redirecting its hard-coded buffer_tank(6)), under the bench's physical policy and constant actions, on every call that
    starts from a state not within 50 % of a Monod pole (conftest.valid_calls)."""
    from gym_sbr2_amd import _capi
    worst = {}
    for name in ["const_2_5", "random_b", "zeros"] + SCENARIO_EPISODES:
        e = golden("sbros_" + name)
        ncall = min(int(e["n_calls"]), valid_calls(e) + 1)
        calls = [k for k in range(1, ncall - 1) if e["step_n_intervals"][k] == 1]
        n = len(calls)
        x = np.zeros((_capi.NX, n)); ctrl = np.zeros((_capi.NCTRL, n))
        for j, k in enumerate(calls):
            p = k - 1
            i0 = np.where(e["iv_call"] == k)[0][0]
            x[:, j] = e["step_x_end"][p]
            ctrl[_capi.C_T, j] = e["step_t"][p]
            ctrl[_capi.C_SO_M1, j], ctrl[_capi.C_SO_M2, j] = e["step_So_m1"][p], e["step_So_m2"][p]
            ctrl[_capi.C_SNO_M1, j], ctrl[_capi.C_SNO_M2, j] = e["step_Sno_m1"][p], e["step_Sno_m2"][p]
            ctrl[_capi.C_IE_DO, j], ctrl[_capi.C_IE_EC, j] = e["step_ie_DO"][p], e["step_ie_EC"][p]
            ctrl[_capi.C_EC_LAST, j] = e["step_EC"][p]
            ctrl[_capi.C_KLA_HIST0:_capi.C_KLA_HIST0 + 10, j] = ([0.0] * 10 + e["iv_Kla"][:i0].tolist())[-10:]
        env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
        env.set_state(x, ctrl)
        acts = e["actions"][calls]
        o, s, r, d = env.step(torch.from_numpy(acts).cuda())
        x1, c1 = env.get_state()
        x1, c1 = _np(x1).T, _np(c1)
        ref_x = e["step_x_end"][calls]
        g = gate(x1, ref_x)
        assert g.max() <= 1.0, (name, g.max())                       # oracle: worst 0.24 (So at aeration switch-on)
        worst[name] = float(g.max())
        # controller outputs are computed before the integration: they equal the reference's to rounding
        assert np.abs(c1[_capi.C_KLA_LAST] - e["step_Kla"][calls]).max() < 1e-10       # Kla up to 240
        assert np.abs(c1[_capi.C_EC_LAST] - e["step_EC"][calls]).max() < 1e-16         # EC up to 5e-4
        assert np.abs(c1[_capi.C_IE_DO] - e["step_ie_DO"][calls]).max() < 1e-16
        assert np.abs(c1[_capi.C_IE_EC] - e["step_ie_EC"][calls]).max() < 1e-16
        # reward from the device's end state: <= 2e-5*S/473 (S = EQI2^2 + OCI^2 <= ~5) for a state inside the gate
        assert np.abs(_np(r) - e["step_reward"][calls]).max() < 5e-7
        # observations inherit the state gate divided by their normaliser (conftest.obs_tolerance)
        ref_o = np.c_[e["step_obs_DO"][calls], e["step_obs_EC"][calls]]
        assert np.all(np.abs(_np(o) - ref_o) <= obs_tolerance(ref_x)), np.abs(_np(o) - ref_o).max()
        env.close()
    print("[info] open loop, device vs reference, worst gate per episode: " + ", ".join("%s %.3f" % kv for kv in worst.items()))
    assert max(worst.values()) <= 0.6                                # oracle: 0.51 (random_b); 0.34 on the scenario episodes


def test_scenario_episodes_closed_loop_against_the_reference_at_tight_tolerance(G, tables):
    """The bench's own workload pinned to the reference: the 18 scenario episodes as one batch, chained
    over the whole episode on the device, against the unmodified reference run with odeint at rtol = atol = 1e-12
    (sbros_scn*_tight.npz) and against the oracle.  Inside the gate on every call on which parity is defined - all 463 calls,
    terminal phases, return and wastage included, for every episode on the bench's scenarios 4..7 under the physical policy -
    and up to the call at which an episode comes near a pole elsewhere; from there on the device must still equal the oracle
    in lockstep (same arithmetic, same garbage) and raise the library's NEAR_POLE flag."""
    full = _closed_loop_batch_against_the_reference(G, tables, SCENARIO_EPISODES, default_tol_bar=1.0)
    assert all(("scn%d_phys" % s) in full for s in BENCH_SCENARIOS) and len(full) == 9


def _closed_loop_batch_against_the_reference(G, tables, names, default_tol_bar, tight_bar=1.0, reward_bar=5e-7):
    """`names` as ONE batch on the device, chained over the whole episode, in lockstep with the oracle, against the reference at
    1e-12 (bar: tight_bar of the gate) and at its default tolerance (default_tol_bar; None = not asserted).  Returns the episodes
    that stayed inside the domain of parity for all 463 calls."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    E = [golden("sbros_" + nm) for nm in names]
    T = [golden("sbros_%s_tight" % nm) for nm in names]
    n, ncall = len(E), 463
    scen = np.array([int(e["scenario"]) for e in E], dtype=np.int32)
    rnd = np.stack([e["rnd"] for e in E])
    nv = np.array([min(valid_calls(e), valid_calls(t)) for e, t in zip(E, T)])
    acts = np.stack([e["actions"][:ncall] for e in E], axis=1).astype(np.float64)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    ora = O.OracleBatch(n)
    obs0 = _np(env.reset(scenario=scen, rnd=rnd)).copy()
    ora.reset(ora.mix(means, stds, scen, rnd))
    x, _ = env.get_state()
    # fill phase vs the reference, every scenario: RK4 x 252 under either scheme, measured 0.003 of the gate (the bound was 1.0 while
    # round 5 tried an adaptive fill phase and stayed there after that was taken back: VERDICT r5 weak 2)
    assert gate(_np(x).T, np.stack([e["x_postfill"] for e in E])).max() < 0.01
    for i, e in enumerate(E):
        assert np.abs(obs0[i] - np.r_[e["reset_obs_DO"], e["reset_obs_EC"]]).max() < 1e-6        # measured 2e-8
    worst_tight, worst_gold = np.zeros(n), np.zeros(n)
    xd, cd = env.get_state()
    for c in range(ncall):
        # lockstep with the oracle: re-synchronised to the device's own state before every call, so that the comparison stays
        # rounding-level also where the closed loop amplifies differences without bound (near a pole)
        ora.load_state(_np(xd), _np(cd))
        o, s, r, d = env.step(torch.from_numpy(acts[c]).cuda())
        ora.step(acts[c])
        xd, cd = env.get_state()
        x, ctrl = _np(xd).T, _np(cd)
        gl = gate(x, ora.envs["x"]).max(axis=1)
        live = c < nv
        assert gl[live].max() < 1e-6, (c, gl.max())
        assert np.isfinite(x).all()
        assert np.array_equal(ctrl[_capi.C_STATUS], ora.envs["status"]), c                # flags agree exactly
        _plans_agree(ctrl, ora, where=live)                                               # ... and so does the plan of every interval
        assert np.array_equal(_np(d), [int(e["step_done"][c]) for e in E])
        assert np.array_equal(ctrl[_capi.C_T], [e["step_t"][c] for e in E])
        if c < ncall - 1:
            for i in np.where(live)[0]:
                worst_tight[i] = max(worst_tight[i], gate(x[i], T[i]["step_x_end"][c]).max())
                worst_gold[i] = max(worst_gold[i], gate(x[i], E[i]["step_x_end"][c]).max())
                assert abs(_np(r)[i] - T[i]["step_reward"][c]) < reward_bar, (names[i], c)
    print("[info] device vs reference(1e-12), worst gate: " + ", ".join("%s %.3f" % (nm, w) for nm, w in zip(names, worst_tight)))
    for i, nm in enumerate(names):
        assert worst_tight[i] <= tight_bar, (nm, worst_tight[i])
        assert default_tol_bar is None or worst_gold[i] <= default_tol_bar, (nm, worst_gold[i])
        if nv[i] == ncall:                                            # the whole episode is inside the domain of parity
            t = T[i]
            assert gate(x[i], t["term_x_after_idle"]).max() <= 1.0, nm
            assert abs(ctrl[_capi.C_RETURN][i] / float(t["episode_return"]) - 1) < 1e-5
            assert abs(ctrl[_capi.C_QW][i] / float(t["term_Qw"]) - 1) < 1e-5
            near = int(ctrl[_capi.C_STATUS][i]) & _capi.ST_NEAR_POLE
            assert not near, nm
        else:
            assert int(ctrl[_capi.C_STATUS][i]) & _capi.ST_NEAR_POLE, nm          # the library says so itself
    env.close()
    return [nm for i, nm in enumerate(names) if nv[i] == ncall]


def test_heldout_episodes_on_the_device(G, tables):
    """the ten HELD-OUT reference episodes (excluded from any fitting of the plan thresholds: the reference's
    own random-walk action model, set-points held 20 calls, a sinusoidal DO set-point through the oxygen knee; new influent
    seeds) chained on the device in lockstep with the oracle - plan equality on every call included - inside 0.6 of the gate of
    the reference at 1e-12 (oracle: 0.373, Ss of `ho_walk_s5`).  Against the reference's DEFAULT tolerance nothing is asserted
    here: its own default run of the walk episodes is 1.3 .. 13 gates from its own 1e-12 run
    (tests/test_oracle_golden.py::test_heldout_walk_episodes_and_the_reference_own_integrator_noise)."""
    # rewards: 5e-6 - the walk episodes keep the dosing PID unsaturated, where EC carries 100 x the deviation of Sno into the cost term
    full = _closed_loop_batch_against_the_reference(G, tables, HELDOUT_EPISODES, default_tol_bar=None, tight_bar=0.6, reward_bar=5e-6)
    assert len(full) == 9 and "ho_held20_s1" not in full


@pytest.mark.parametrize("scheme", [0, 1])
def test_config2_4096_envs_full_episode_against_oracle(G, tables, scheme):
    """BASELINE.json configs[1]: 4096 envs, fixed-step RK4 (cfg.scheme = 0, as the config words it; and the library's default
    scheme 1), deterministic influent (rnd = 0), scenario = env id mod 8, seeded random float32 actions; EVERY env on EVERY call
    (1.9 M env-calls), against the C oracle.

    Policy: even envs draw U[0,8] x U[0,15] every call (the aggressive case: both clamps, bang-bang dosing); odd envs
    draw the DO set-point from U[0,2.5].  The reference model has no guards and the aggressive policy drives
    ammonia negative and towards the pole of Snh/(Knh+Snh) in most envs (status bit NEAR_POLE); there no two fp64
    computations agree (control: two CPU builds of the oracle differ by up to 1e4 gates), so:
      lockstep  - before each call the oracle is re-synchronised to the device's own state, then both take the call;
                  envs that are well-posed before and after the call must agree TIGHTLY, flags must agree exactly
      free run  - a second oracle runs the whole episode on its own; envs never flagged NEAR_POLE on either side
                  must stay within 1e-6 of the gate (CPU control: <= 3e-10), all envs must stay finite."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n, ncall = 4096, 463
    rs = np.random.RandomState(2)
    scen = (np.arange(n) % 8).astype(np.int32)
    rnd = np.zeros((n, 48))
    cfg = _capi.default_config(); cfg.scheme = scheme
    env = G.SbrOSVec(n, out_dtype=torch.float32, config=cfg)
    sync = O.OracleBatch(n, O.default_params(scheme=scheme), nthreads=8)
    free = O.OracleBatch(n, O.default_params(scheme=scheme), nthreads=8)
    obs = _np(env.reset(scenario=scen, rnd=rnd))
    infl = free.mix(means, stds, scen, rnd)
    oobs = free.reset(infl); sync.reset(infl)
    assert np.abs(obs - oobs).max() < 1e-5                            # float32 outputs
    x, ctrl = env.get_state()
    assert gate(_np(x).T, free.envs["x"]).max() < 1e-6                # post-fill (252 substeps, open loop)
    ret = np.zeros(n); free_gate = []; worst_sync_ok = 0.0; worst_sync_flagged = 0.0
    odd = (np.arange(n) % 2 == 1)
    for c in range(ncall):
        a = np.column_stack([rs.uniform(0, 8, n), rs.uniform(0, 15, n)])
        a[odd, 0] = rs.uniform(0, 2.5, odd.sum())
        a = a.astype(np.float32)
        xb, cb = _np(x), _np(ctrl)
        sync.load_state(xb, cb)
        pole_before = (cb[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) != 0
        o, s, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = sync.step(a.astype(np.float64))
        _, _, frr, fod = free.step(a.astype(np.float64), want_obs=False)
        x, ctrl = env.get_state()
        xn, cn = _np(x).T, _np(ctrl)
        # --- lockstep: one call from identical state
        assert np.array_equal(_np(d), od) and np.array_equal(od, fod), c
        assert np.array_equal(cn[_capi.C_STATUS], sync.envs["status"]), c                 # flags agree exactly
        ok = ~pole_before & ((cn[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) == 0)
        plans = _plans_agree(cn, sync, where=ok)                                          # from identical states: the same plan
        assert np.all(plans == 0) if scheme == 0 else np.all((plans & 127) >= 1)
        gs = gate(xn, sync.envs["x"]).max(axis=1)
        worst_sync_ok = max(worst_sync_ok, gs[ok].max())
        worst_sync_flagged = max(worst_sync_flagged, gs[~ok].max(initial=0.0))
        assert gs[ok].max() < 1e-6, (c, gs[ok].max())                 # = 1e-11 relative (464 idle substeps on the last call)
        assert np.isfinite(xn).all()
        for got, ref, rt, at in ((_np(o), oo, 2e-7, 1e-7), (_np(s), os_, 2e-7, 1e-7)):    # float32 cast of the outputs
            assert np.allclose(got[ok], ref[ok], rtol=rt, atol=at)
        assert np.allclose(_np(r)[ok], orr[ok], rtol=2e-7, atol=1e-10)
        assert np.array_equal(cn[_capi.C_T], sync.envs["t"])
        assert np.abs(cn[_capi.C_KLA_LAST] - sync.envs["kla_last"]).max() < 1e-9           # computed before the integration:
        assert np.abs(cn[_capi.C_EC_LAST] - sync.envs["ec_last"]).max() < 1e-15            # tight for every env
        assert np.abs(cn[_capi.C_IE_DO] - sync.envs["ie_do"]).max() < 1e-15 and np.abs(cn[_capi.C_IE_EC] - sync.envs["ie_ec"]).max() < 1e-15
        assert np.abs(cn[_capi.C_RETURN] - sync.envs["ret"])[ok].max() < 1e-12
        # --- free run
        if c < ncall - 1:
            free_gate.append(gate(xn, free.envs["x"]).max(axis=1))
        ret += frr
    free_gate = np.array(free_gate)
    st_dev = cn[_capi.C_STATUS].astype(int); st_free = free.envs["status"].astype(int)
    clean = ((st_dev | st_free) & _capi.ST_NEAR_POLE) == 0
    print("lockstep worst gate: well-posed %.3e, flagged %.3e | free run: %d of %d envs never near a pole, their worst gate "
          "%.3e; all envs median %.3e p99 %.3e max %.3e; negative-concentration flag on %d envs" % (
              worst_sync_ok, worst_sync_flagged, clean.sum(), n, free_gate[:, clean].max(), np.median(free_gate),
              np.percentile(free_gate, 99), free_gate.max(), ((st_dev & _capi.ST_NEGATIVE) != 0).sum()))
    assert clean.sum() > 1000                                         # the comparison below is not vacuous
    assert free_gate[:, clean].max() < 1e-6
    assert np.median(free_gate) < 1e-7
    assert np.all(_np(d) == 1) and (st_dev & _capi.ST_NONFINITE).sum() == 0
    dret = np.abs(_np(ctrl[_capi.C_RETURN]) - ret)
    assert dret[clean].max() < 1e-10 and np.median(dret) < 1e-13
    st = env.stats(ctrl[_capi.C_RETURN])
    got = _np(ctrl[_capi.C_RETURN])
    assert st["count"] == n and abs(st["sum"] - got.sum()) < 1e-9 * abs(got.sum())
    assert st["min"] == got.min() and st["max"] == got.max()
    env.close()


def test_two_waves_per_simd_kernel_build_matches_oracle(G, tables):
    """Batches above SBR_SMALL_BATCH = 49152 envs run the 256-thread-workgroup instantiation of k_step (one source, two
    workgroup sizes), and from 65536 envs up more than one wave is resident per SIMD.  Lockstep check of that regime: 98368
    envs (not a multiple of 256: the last workgroup is ragged), the oracle follows a 512-env sample (first, middle and last
    waves) for 60 calls, crossing the anoxic -> aerobic boundary (one double step)."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n, calls = 98304 + 64, 60
    pick = np.r_[0:192, n // 2:n // 2 + 128, n - 192:n]
    scen = (np.arange(n) % 8).astype(np.int32)
    env = G.SbrOSVec(n, out_dtype=torch.float64)
    env.reset(seed=3, scenario=scen)
    ora = O.OracleBatch(len(pick))
    rs = np.random.RandomState(5)
    x, ctrl = env.get_state()
    worst = 0.0
    for c in range(calls):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)]).astype(np.float32)
        ora.load_state(_np(x)[:, pick], _np(ctrl)[:, pick])
        o, s, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = ora.step(a[pick].astype(np.float64))
        x, ctrl = env.get_state()
        g = gate(_np(x).T[pick], ora.envs["x"]).max()
        worst = max(worst, g)
        assert g < 1e-6 and np.array_equal(_np(d)[pick], od)
        _plans_agree(_np(ctrl), ora, pick=pick)            # the parked build carries the plan through an LDS slot
        assert np.abs(_np(o)[pick] - oo).max() < 1e-10 and np.abs(_np(r)[pick] - orr).max() < 1e-12
        assert np.array_equal(_np(ctrl)[_capi.C_T, pick], ora.envs["t"])
    assert int(_np(ctrl)[_capi.C_STEPS].min()) == calls and np.isfinite(_np(x)).all()
    print("two-wave build, lockstep worst gate %.3e" % worst)
    env.close()


@pytest.mark.parametrize("n", [1, 63, 65, 130])
def test_ragged_batch_sizes(G, tables, n):
    """Batches that do not fill their last wave (every other test uses multiples of 64): reset, 70 calls across the first
    phase boundary, rollout, export/import round trip - against the oracle, and nothing written out of bounds."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    scen = (np.arange(n) % 8).astype(np.int32)
    rs = np.random.RandomState(n)
    z = rs.randn(n, 48)
    env = G.SbrOSVec(n, out_dtype=torch.float32)
    ora = O.OracleBatch(n)
    guard = torch.full((n + 64, 18), -7.0, device="cuda")          # obs buffer with a canary tail
    env.obs = guard[:n]
    for bad in (guard[:n + 1], guard[:n].double(), guard[:n].cpu(), guard[:n, ::2]):     # replaced output buffers are validated
        with pytest.raises(ValueError):
            env.obs = bad
    assert env.obs.data_ptr() == guard.data_ptr()
    obs = _np(env.reset(scenario=scen, rnd=z)); oobs = ora.reset(ora.mix(means, stds, scen, z))
    assert np.abs(obs - oobs).max() < 1e-5 and bool((guard[n:] == -7.0).all())
    for c in range(70):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)]).astype(np.float32)
        o, s, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = ora.step(a.astype(np.float64))
        assert np.allclose(_np(o), oo, rtol=2e-6, atol=1e-6) and np.allclose(_np(r), orr, rtol=2e-6, atol=1e-9)
        assert np.array_equal(_np(d), od)
    assert bool((guard[n:] == -7.0).all())
    x, ctrl = env.get_state()
    assert gate(_np(x).T, ora.envs["x"]).max() < 1e-6 and np.array_equal(_np(ctrl)[_capi.C_STEPS], np.full(n, 70.0))
    env.set_state(x, ctrl)                                           # export -> import is the identity
    x2, ctrl2 = env.get_state()
    assert torch.equal(x, x2) and torch.equal(ctrl, ctrl2)
    ret = env.rollout(30, policy_seed=4)
    assert np.abs(_np(ret) - ora.rollout(30, 4)).max() < 1e-9 and _np(ret).shape == (n,)
    env.close()


def test_g2anet_reward_option(G, tables):
    """cfg.reward_kind = 1: same plant, the piecewise-linear reward of module_reward_continuous_G2ANET.py.
    The device reward of every call must equal the reference function evaluated on the device's own end state."""
    from gym_sbr2_amd import _capi
    k = golden("reward_g2anet_kat")
    cfg = _capi.default_config(); cfg.reward_kind = 1
    n = len(k["X"])
    env = G.SbrOSVec(n, out_dtype=torch.float64, config=cfg)
    p = O.default_params(); p.reward_kind = 1
    ora = O.OracleBatch(n, params=p)
    means, stds = tables
    z = np.random.RandomState(1).randn(n, 48); scen = (np.arange(n) % 8).astype(np.int32)
    env.reset(scenario=scen, rnd=z); ora.reset(ora.mix(means, stds, scen, z))
    a = np.column_stack([np.full(n, 2.0), np.full(n, 5.0)]).astype(np.float32)
    for c in range(60):
        _, _, r, _ = env.step(torch.from_numpy(a).cuda()); _, _, orr, _ = ora.step(a.astype(np.float64))
        assert np.abs(_np(r) - orr).max() < 1e-12
    env.close()
    # known answers: inject the fixture's states (they straddle every kink of the reward; a few sit ON a Monod pole,
    # e.g. Snh = -1, and turn NaN within one interval - there the device and the reference function must both be NaN)
    cfg.terminal = 0                                   # no settle/draw/idle: the inspected state is the reward's state
    env = G.SbrOSVec(n, out_dtype=torch.float64, config=cfg)
    x = k["X"].T.copy(); ctrl = np.zeros((_capi.NCTRL, n)); ctrl[_capi.C_T] = 0.45
    env.set_state(x, ctrl)
    import ctypes as C
    fn = O.lib().sbro_reward_g2anet; fn.restype = C.c_double
    _, _, r, _ = env.step(torch.zeros(n, 2))
    x1, _ = env.get_state()
    want = np.array([fn(O._p(np.ascontiguousarray(v))) for v in _np(x1).T])
    finite = np.isfinite(want)
    assert np.array_equal(np.isfinite(_np(r)), finite) and finite.sum() >= 90
    assert np.abs(_np(r)[finite] - want[finite]).max() < 1e-14
    assert (_np(env.status())[~finite] & _capi.ST_NONFINITE).all()      # and the status flag says so
    env.close()


def test_oci_reward_option(G, tables):
    """cfg.reward_kind = 2: the operating-cost reward of module_reward_continuous.py:4-65 on the SBROS-v1
    plant.  Whole episodes against the oracle (whose reward function is pinned bit-exactly by the reference's own, see
    tests/test_oracle_golden.py): the per-call reward, the running sum(Kla) row, the end-of-cycle reward of the done call
    (sum over 252 + 466 + 1 list entries, pumping of Qw and Qeff, ammonia penalty), step path and fused rollout."""
    from gym_sbr2_amd import _capi
    cfg = _capi.default_config(); cfg.reward_kind = 2; cfg.act_f64 = 1
    n = 192
    env = G.SbrOSVec(n, out_dtype=torch.float64, config=cfg)
    p = O.default_params(); p.reward_kind = 2
    ora = O.OracleBatch(n, params=p)
    means, stds = tables
    z = np.random.RandomState(2).randn(n, 48); scen = (np.arange(n) % 8).astype(np.int32)
    env.reset(scenario=scen, rnd=z); ora.reset(ora.mix(means, stds, scen, z))
    assert np.array_equal(_np(env.ctrl_row(_capi.C_KLA_SUM)), ora.envs["kla_sum"])          # 126 * k_fill, summed in order
    a = np.column_stack([np.linspace(0.5, 4.0, n), np.linspace(0.0, 12.0, n)])          # a spread of set-points
    at = torch.from_numpy(a).cuda()
    for c in range(463):
        _, _, r, d = env.step(at); _, _, orr, od = ora.step(a)
        assert np.array_equal(_np(d), od)
        if c < 462:
            assert np.abs(_np(r) - orr).max() < 1e-10, c
    assert od.all()
    # The done call's reward is a function of Qw and sum(Kla), both products of 463 free-running closed-loop calls.  With
    # their differences removed the two rewards agree to rounding on EVERY env (measured 1.9e-14), i.e. the formula is the
    # same; the differences themselves are bounded on the envs that stayed away from a Monod pole (measured there:
    # sum(Kla) 2.6e-8 relative, Qw 1.4e-10, reward 5e-10; 134 of these 192 set-point pairs do reach a pole).
    ks, oks = _np(env.ctrl_row(_capi.C_KLA_SUM)), ora.envs["kla_sum"]
    dqw = _np(env.ctrl_row(_capi.C_QW)) - ora.envs["qw"]
    coef = 8.000000000006622 / 1800 * 1.32 * (0.002 / 24)
    assert np.abs((_np(r) - orr) + 0.05 * dqw + coef * (ks - oks)).max() < 1e-12
    clean = ((_np(env.status()) | ora.envs["status"].astype(int)) & _capi.ST_NEAR_POLE) == 0
    assert clean.sum() >= 40 and oks.min() > 1000.0
    assert np.abs(ks / oks - 1)[clean].max() < 1e-6 and np.abs(dqw)[clean].max() < 1e-8
    assert np.abs(_np(r) - orr)[clean].max() < 1e-8
    rr = _np(r)
    assert (rr < -200).any() and (rr > 0).any()            # both sides of the ammonia penalty occur in this batch
    x, ctrl = env.get_state()
    want = 0.5 - (8.000000000006622 / 1800 * (1.32 * ks * (0.002 / 24)) + 0.05 * _np(ctrl)[_capi.C_QW] + 0.004 * 0.66)
    assert np.abs(np.where(rr < -200, rr + 246, rr) - want).max() < 1e-12
    # the row survives a get/set round trip, and the fused rollout keeps it too
    env.reset(scenario=scen, rnd=z); ora.reset(ora.mix(means, stds, scen, z))
    ret = env.rollout(463, 9); oret = ora.rollout(463, 9)
    ok = np.isfinite(oret) & (((_np(env.status()) | ora.envs["status"].astype(int)) & _capi.ST_NEAR_POLE) == 0)
    assert ok.sum() >= 20 and np.abs(_np(ret)[ok] - oret[ok]).max() < 1e-7      # return ~ 231: 4e-10 relative
    x, ctrl = env.get_state(); env.set_state(x, ctrl)
    assert np.array_equal(_np(env.ctrl_row(_capi.C_KLA_SUM)), _np(ctrl)[_capi.C_KLA_SUM])
    env.close()


def test_step_is_hip_graph_capturable(G):
    """sbr_step neither allocates nor synchronises, so a sequence of steps can be captured in a HIP graph and replayed;
    the replayed plant must be bit-identical to stepping eagerly."""
    n, k = 4096, 64
    scen = (np.arange(n) % 8).astype(np.int32)
    acts = torch.rand(k, n, 2, device="cuda") * torch.tensor([2.5, 15.0], device="cuda")
    a_env, b_env = G.SbrOSVec(n), G.SbrOSVec(n)
    a_env.reset(seed=2, scenario=scen); b_env.reset(seed=2, scenario=scen)
    a_env.step(acts[0]); b_env.step(acts[0])                                   # warm up (module load) outside the capture
    a_env.reset(seed=2, scenario=scen); b_env.reset(seed=2, scenario=scen)
    torch.cuda.synchronize()
    graph = a_env.capture_steps([acts[j] for j in range(k)])                  # capturing does not execute
    for _ in range(3):
        graph.replay()
    for _ in range(3):
        for j in range(k):
            b_env.step(acts[j])
    torch.cuda.synchronize()
    xa, ca = a_env.get_state(); xb, cb = b_env.get_state()
    assert torch.equal(xa, xb) and torch.equal(ca, cb) and int(ca[20].max().item()) == 3 * k
    assert torch.equal(a_env.obs, b_env.obs) and torch.equal(a_env.reward, b_env.reward)
    a_env.close(); b_env.close()


def test_status_flags_report_leaving_the_physical_domain(G):
    """The reference silently returns garbage once a concentration is driven to a Monod pole; the library reproduces
    the numbers but raises sticky flags.  States are injected with set_state and one call is taken."""
    from gym_sbr2_amd import _capi
    e = golden("sbros_const_2_5")
    n = 5
    x = np.tile(e["step_x_end"][300][:, None], (1, n))
    ctrl = np.zeros((_capi.NCTRL, n))
    ctrl[_capi.C_T] = e["step_t"][300]
    ctrl[_capi.C_SO_M1] = ctrl[_capi.C_SO_M2] = x[8, 0]
    ctrl[_capi.C_SNO_M1] = ctrl[_capi.C_SNO_M2] = x[9, 0]
    x[10, 1] = -0.1        # ammonia slightly negative: NEGATIVE only
    x[10, 2] = -0.7        # within 50 % of the pole at -Knh = -1: NEAR_POLE too
    x[9, 3] = -0.3         # nitrate beyond -Kno/2 at the START of the interval only: flags look at the END state, and a
                           # negative Monod term reverses denitrification and pulls Sno back up within the interval
    x[5, 4] = float("nan")
    env = G.SbrOSVec(n, out_dtype=torch.float64)
    env.set_state(x, ctrl)
    env.step(torch.zeros(n, 2))
    st = _np(env.status()).tolist()
    assert st[0] == 0
    assert st[1] == _capi.ST_NEGATIVE
    assert st[2] == _capi.ST_NEGATIVE | _capi.ST_NEAR_POLE
    assert (st[3] & _capi.ST_NONFINITE) == 0                          # fast variable: recovers, whatever the other bits say
    assert st[4] & _capi.ST_NONFINITE
    env.step(torch.zeros(n, 2))
    assert _np(env.status()).tolist()[1:3] == st[1:3]                 # sticky
    env.reset(rnd=np.zeros((n, 48)))
    assert _np(env.status()).tolist() == [0] * n                      # cleared by reset
    env.close()
    # Round 6: a state OUTSIDE the model's domain does not raise the step count.  The knee's count grows with the oxygen rate at
    # So = 0 - up to 64 steps - because inside the domain that rate is bounded by the Monod factors of Ss and Snh; with Ss beyond its
    # pole (Ss < -K_S: factor 21) or a NaN the "rate" is garbage, and ONE such lane at 64 steps made a 65 536-env launch last 60 us
    # instead of 11 (the uniform policy's late calls; its done call 3 ms).  Aerobic interval from So = 0 (the knee), Kla > 0:
    m = 6
    i = int(np.where(e["iv_kind"] == 1)[0][0])
    x = np.tile(e["iv_x_start"][i][:, None], (1, m))
    ctrl = np.zeros((_capi.NCTRL, m))
    ctrl[_capi.C_T] = e["iv_t_start"][i]
    ctrl[_capi.C_SO_M1] = ctrl[_capi.C_SO_M2] = x[8, 0]
    ctrl[_capi.C_SNO_M1] = ctrl[_capi.C_SNO_M2] = x[9, 0]
    x[5, 1] *= 3.0; x[6, 1] *= 3.0       # in the domain, three times the biomass: the stability rule asks for more than four steps
    x[5, 2] *= 1e6                       # in the domain (absurd biomass): the cap of 64
    x[2, 3] = -10.5                      # Ss beyond its pole at -K_S = -10: Monod factor 21
    x[10, 4] = float("nan")              # NaN ammonia
    x[10, 5] = -1.2                      # Snh beyond its pole at -K_NH = -1: factor 6
    env = G.SbrOSVec(m, out_dtype=torch.float64, action_dtype=torch.float64)
    env.set_state(x, ctrl)
    env.step(torch.tensor([[2.0, 5.0]] * m, dtype=torch.float64))
    steps = (_np(env.plan()) & 127).tolist()
    assert steps[0] == 4 and 5 <= steps[1] <= 14 and steps[2] == 64 and steps[3:] == [4, 4, 4], steps
    ora = O.OracleBatch(m)
    ora.load_state(x, ctrl)
    ora.step(np.tile([2.0, 5.0], (m, 1)))
    assert (ora.envs["scheme_plan"] & 127).tolist() == steps          # the oracle applies the same rule
    env.close()


def test_fused_rollout_equals_step_by_step_and_oracle(G, tables):
    """BASELINE.json configs[4]: on-GPU random policy, fused action-sample + step.  The fused kernel must
    give the same plant as replaying its sampled actions through sbr_step, and the oracle's Philox policy."""
    means, stds = tables
    n, steps = 512, 463
    scen = (np.arange(n) % 8).astype(np.int32)
    z = np.random.RandomState(3).randn(n, 48)
    a_env = G.SbrOSVec(n, out_dtype=torch.float64, first_env_id=1000)
    b_env = G.SbrOSVec(n, out_dtype=torch.float64, first_env_id=1000)
    a_env.reset(scenario=scen, rnd=z); b_env.reset(scenario=scen, rnd=z)
    ret, acts = a_env.rollout(steps, policy_seed=11, return_actions=True)
    tot = torch.zeros(n, dtype=torch.float64, device="cuda")
    for c in range(steps):
        _, _, r, _ = b_env.step(acts[c])
        tot += r
    xa, ca = a_env.get_state(); xb, cb = b_env.get_state()
    # k_rollout and k_step inline the same device functions but are separate kernels (FMA contraction may differ)
    from gym_sbr2_amd import _capi
    keep = [r_ for r_ in range(_capi.NCTRL) if r_ != _capi.C_PLAN]
    assert gate(_np(xa).T, _np(xb).T).max() < 1e-6 and torch.allclose(ca[keep], cb[keep], rtol=1e-11, atol=1e-13)
    # the plan row is sbr_step's report (its last interval: the aerobic one of the done call); the fused rollout reports none
    assert bool((ca[_capi.C_PLAN] == 0).all()) and bool(((cb[_capi.C_PLAN].to(torch.int64) & 127) >= 1).all())
    for row in (_capi.C_T, _capi.C_DONE, _capi.C_STEPS, _capi.C_STATUS):
        assert torch.equal(ca[row], cb[row])
    assert torch.allclose(ret, tot, rtol=0, atol=1e-12)
    ora = O.OracleBatch(n, nthreads=8, first_env_id=1000)
    ora.reset(ora.mix(means, stds, scen, z))
    assert np.array_equal(_np(acts[:5]), ora.policy_actions(5, 11))  # same Philox stream, same float32 actions
    oret = ora.rollout(steps, 11)
    # free-running closed loop under the uniform random policy: tight on envs never flagged NEAR_POLE (see the 4096-env test)
    clean = ((_np(ca[_capi.C_STATUS]).astype(int) | ora.envs["status"].astype(int)) & _capi.ST_NEAR_POLE) == 0
    dret = np.abs(_np(ret) - oret)
    gx = gate(_np(xa).T, ora.envs["x"]).max(axis=1)
    print("rollout vs oracle: %d of %d envs never near a pole; their worst |d return| %.2e, worst gate %.2e; all: median gate %.2e"
          % (clean.sum(), n, dret[clean].max(initial=0), gx[clean].max(initial=0), np.median(gx)))
    assert clean.sum() >= 10 and dret[clean].max() < 1e-10 and gx[clean].max() < 1e-6
    assert np.median(dret) < 1e-13 and np.median(gx) < 1e-7 and np.isfinite(_np(xa)).all()
    # split rollouts continue the same action stream (the counter is the call index since reset)
    b_env.reset(scenario=scen, rnd=z)
    r1 = b_env.rollout(200, policy_seed=11); r2 = b_env.rollout(steps - 200, policy_seed=11)
    assert torch.allclose(r1 + r2, ret, rtol=0, atol=1e-12)
    a_env.close(); b_env.close()


def test_size_independent_properties_at_65536(G):
    """BASELINE.json configs[2] size (65536 envs, stochastic influent): properties that need no oracle."""
    from gym_sbr2_amd import _capi
    n = 65536
    env = G.SbrOSVec(n)
    scen = (np.arange(n) % 8).astype(np.int32)
    env.reset(seed=5, scenario=scen)
    # seeded (the one failure this test ever had was on an input nobody could name afterwards); another seed:
    # SBR_TEST_SEED=<n>.  The seed is printed, and so are the offending envs if a property fails.
    seed = int(os.environ.get("SBR_TEST_SEED", "20261005"))
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    print("[info] test_size_independent_properties_at_65536: action seed %d" % seed)
    acts = torch.rand(463, n, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")
    # replicas: envs 0..7 re-run in a small handle with the same global ids give bit-identical plants
    small = G.SbrOSVec(64, first_env_id=0)
    small.reset(seed=5, scenario=scen[:64])
    v0 = None
    for c in range(463):
        o, s, r, d = env.step(acts[c])
        small.step(acts[c, :64].contiguous())
        if c == 0:
            first = env.get_state()[0].clone()
            v0 = first[0]
        if c < 462:
            assert int(d.sum().item()) == 0
    assert int(d.sum().item()) == n                                   # every env finishes on call 463
    x, ctrl = env.get_state(); xs, cs = small.get_state()
    assert torch.equal(x[:, :64], xs) and torch.equal(ctrl[:, :64], cs)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(ctrl[_capi.C_RETURN]).all())
    # volume only grows by the dosed carbon: V after one interval in [V0, V0 + EC_max * t_delta]
    cfg = env.cfg
    assert bool(((v0 >= cfg.WV - 1e-12) & (v0 <= cfg.WV + cfg.EC_max * cfg.t_delta * 2 + 1e-12)).all())
    # inert soluble Si is never produced: after the fill it can only be diluted by the dosed carbon
    assert bool((x[1] <= first[1] * (1 + 1e-12)).all()) and bool((x[1] > 0).all())
    # effluent + waste sludge were drawn - for every env whose plant stayed inside the model's domain.  The uniform policy of this
    # test drives ~86 % of the envs near a Monod pole, where the reference model's numbers have no physical meaning.  What then
    # happens in the draw was CAPTURED in round 6 (scripts/analysis/draw_sweep.py on the CPU oracle: 9 instances in
    # 12 M envs; draw_sweep_gpu.py on the device: 25 in 60 M, each equal to the oracle's replay; profiles/r06_draw_sweep.json):
    # every instance is an env flagged NEAR_POLE | NEGATIVE long before the done call (So ~ 3e4 g/m3, Snh ~ 8e3), its sludge has
    # collapsed (Xf ~ 400 - 600 instead of ~3 900), the retained layers hold LESS sludge than biomass_setpoint x residual volume,
    # so `waste_sX_weight` (gym_SBR_oneshot.py:2363) is negative and the wastage quotient Qw = waste / (sX[0] - set-point) of
    # :2376 with it: Qw = -0.7 ... -30 m3, and V = V_settled - Qeff - Qw comes out above WV.  The reference's own formula on a
    # state outside its domain, reproduced; not a defect of sbr_draw.  Hence: an env with V >= WV after the done call must be a
    # flagged one AND have a negative Qw, and the volume balance V + Qeff + Qw = the settled volume holds for every env.
    st = ctrl[_capi.C_STATUS].to(torch.int64)
    sane = (st & _capi.ST_NEAR_POLE) == 0
    qw = ctrl[_capi.C_QW]
    odd = x[0] >= cfg.WV
    if bool(odd.any()):
        ids = torch.nonzero(odd).flatten().tolist()
        print("[info] seed %d: V >= WV after the done call on envs %s: V %s Qw %s status %s"
              % (seed, ids, x[0][odd].tolist(), qw[odd].tolist(), st[odd].tolist()))
    assert int(sane.sum().item()) > 1000 and bool((x[0][sane] < cfg.WV).all())
    assert bool(((st[odd] & _capi.ST_NEAR_POLE) != 0).all()) and bool((qw[odd] < 0).all())
    fin = torch.isfinite(x[0]) & torch.isfinite(qw)
    settled = x[0] + cfg.Qeff + qw                        # = the volume before the draw when no whole layer was wasted (true of every
    whole = (settled < cfg.WV - 1e-9) & fin               # sane env: its first retained layer alone covers the wastage)
    assert bool((settled[fin & ~whole] <= cfg.WV + 463 * cfg.EC_max * cfg.t_delta + 1e-9).all()) and not bool((whole & sane).any())
    # masked reset touches only the selected envs
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda"); mask[::2] = 1
    env.reset(seed=6, scenario=scen, mask=mask)
    x2, c2 = env.get_state()
    assert torch.equal(x2[:, 1::2], x[:, 1::2]) and bool((c2[_capi.C_DONE][::2] == 0).all()) and bool((c2[_capi.C_DONE][1::2] == 1).all())
    env.close(); small.close()


def test_bench_workload_parity_at_65536_with_the_physical_policy(G, tables):
    """The workload bench.py times by default (BASELINE.json configs[2]: 65536 envs, stochastic influent, per-call random
    set-points; `--policy physical`: u_DO ~ U[0, 2.5], u_EC ~ U[0, 15], scenarios 4..7) against the oracle, FREE-RUNNING over
    the whole episode on a 512-env sample (first, middle and last wavefronts): the sample is never re-synchronised to the
    device, so 463 calls of differences accumulate.  Under this policy every env stays inside the model's physical domain,
    so the comparison is tight on ALL sampled envs (no env is excused), and the status row says so for all 65536."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 65536
    pick = np.r_[0:192, n // 2:n // 2 + 128, n - 192:n]
    gid = np.arange(n)
    scen = (4 + gid % 4).astype(np.int32)
    env = G.SbrOSVec(n)                                               # float32 actions and outputs, like the bench
    obs = _np(env.reset(seed=1000, scenario=scen)).copy()
    ora = O.OracleBatch(len(pick))
    ora.first_env_id = 0
    z = np.stack([O.OracleBatch(1, first_env_id=int(i)).normals(1000)[0] for i in pick])
    oobs = ora.reset(ora.mix(means, stds, scen[pick], z))
    assert np.abs(obs[pick] - oobs).max() < 1e-5
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    pool = torch.rand(64, n, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
    worst, ret = 0.0, np.zeros(len(pick))
    for c in range(463):
        a = pool[c & 63]
        o, s_, r, d = env.step(a)
        oo, os_, orr, od = ora.step(_np(a)[pick].astype(np.float64))
        ret += orr
        assert np.array_equal(_np(d)[pick], od)
        if c % 20 == 0 or c >= 460:
            x, ctrl = env.get_state()
            g = gate(_np(x).T[pick], ora.envs["x"]).max()
            worst = max(worst, g)
            assert g < 1e-6, (c, g)                                   # free-running, every sampled env
            assert np.allclose(_np(r)[pick], orr, rtol=2e-7, atol=1e-10) and np.abs(_np(o)[pick] - oo).max() < 1e-5
    x, ctrl = env.get_state()
    ctrl = _np(ctrl)
    # all 65536 envs stay clear of the Monod poles and finite; a handful (measured 9 of 65536) dip a concentration below
    # -1e-6 (SBR_ST_NEGATIVE), which the sampled envs - asserted tightly above whatever their flags - may or may not include
    st = ctrl[_capi.C_STATUS].astype(np.int64)
    assert np.all((st & (_capi.ST_NEAR_POLE | _capi.ST_NONFINITE)) == 0) and ((st & _capi.ST_NEGATIVE) != 0).mean() < 1e-3
    assert np.array_equal(st[pick], ora.envs["status"].astype(np.int64))
    assert np.all(ctrl[_capi.C_DONE] == 1) and np.abs(ctrl[_capi.C_RETURN][pick] - ret).max() < 1e-10
    assert np.abs(ctrl[_capi.C_QW][pick] / ora.envs["qw"] - 1).max() < 1e-9
    print("bench workload (physical policy), free-running sample: worst gate %.3e" % worst)
    env.close()


def test_bench_walk_policy_workload_at_65536_against_oracle(G, tables):
    """`bench.py --policy walk` (the reference's own action model, get_available_actions gym_SBR_oneshot.py:440-459: from [0, 15]
    every call moves each set-point by -0.1 / 0 / +0.1 and -5 / 0 / +5 inside [0, 8] x [0, 15]) on the bench's 65536 envs against
    the oracle, free-running over the whole episode on a 512-env sample.  With the aeration set-point walking near 0 a few
    percent of the envs end an episode near a Monod pole (SBR_ST_NEAR_POLE; the reference model has no guards): those are
    excused from the state comparison, as in the configs[1] test, and must carry the same flags on both sides."""
    import bench
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 65536
    pick = np.r_[0:192, n // 2:n // 2 + 128, n - 192:n]
    scen = (4 + np.arange(n) % 4).astype(np.int32)
    env = G.SbrOSVec(n)
    obs = _np(env.reset(seed=1000, scenario=scen)).copy()
    ora = O.OracleBatch(len(pick))
    z = np.stack([O.OracleBatch(1, first_env_id=int(i)).normals(1000)[0] for i in pick])
    oobs = ora.reset(ora.mix(means, stds, scen[pick], z))
    assert np.abs(obs[pick] - oobs).max() < 1e-5
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    cur = torch.tensor([0.0, 15.0], device="cuda").repeat(n, 1)
    ret, moved = np.zeros(len(pick)), 0
    for c in range(463):
        nxt = bench.walk_move(cur, torch.randint(0, 3, (n, 2), device="cuda", generator=gen), torch)
        moved += int((nxt != cur).any(dim=1).sum().item())
        cur = nxt
        assert float(cur.min().item()) >= 0.0 and float(cur[:, 0].max().item()) <= 8.0 and float(cur[:, 1].max().item()) <= 15.0
        o, s_, r, d = env.step(cur)
        oo, os_, orr, od = ora.step(_np(cur)[pick].astype(np.float64))
        ret += orr
        assert np.array_equal(_np(d)[pick], od)
    assert moved > 0.5 * n * 463                                     # the walk walks
    x, ctrl = env.get_state()
    ctrl = _np(ctrl)
    st, ost = ctrl[_capi.C_STATUS].astype(np.int64), ora.envs["status"].astype(np.int64)
    assert np.all((st & _capi.ST_NONFINITE) == 0)
    pole = ((st[pick] | ost) & _capi.ST_NEAR_POLE) != 0
    assert pole.mean() < 0.2 and ((st & _capi.ST_NEAR_POLE) != 0).mean() < 0.1, (pole.mean(), ((st & _capi.ST_NEAR_POLE) != 0).mean())
    ok = ~pole
    assert np.array_equal(st[pick][ok], ost[ok])
    g = gate(_np(x).T[pick][ok], ora.envs["x"][ok]).max()
    assert g < 1e-6, g                                                # free-running over 463 calls, every well-posed sampled env
    assert np.all(ctrl[_capi.C_DONE] == 1) and np.abs(ctrl[_capi.C_RETURN][pick][ok] - ret[ok]).max() < 1e-9
    print("bench workload (walk policy), free-running sample: %d of %d sampled envs well-posed, worst gate %.3e" % (ok.sum(), len(pick), g))
    env.close()


def test_configs3_per_gpu_shape_against_oracle_and_the_unsharded_batch(G, tables):
    """BASELINE.json configs[3] at ITS OWN per-GPU shape: rank 3 of 8 of a 262144-env batch = 32768 envs with global ids
    98304..131071, which run in the 64-thread-workgroup build of k_step (SBR_SMALL_BATCH).  66 calls under the physical policy,
    across the anoxic -> aerobic boundary (a double step), a 512-env oracle sample free-running; and the same global ids
    stepped inside ONE 262144-env handle (256-thread workgroups, two waves per SIMD) give the same returns and states, bit for
    bit: influent noise, scenario and arithmetic are keyed by the global env id, not by the shard."""
    from gym_sbr2_amd import ShardedSbrOS
    means, stds = tables
    n_global, world, rank, calls = 262144, 8, 3, 66
    sh = ShardedSbrOS(n_global, rank=rank, world=world, device=0)
    n, first = sh.stop - sh.start, sh.start
    assert (n, first) == (32768, 98304) and sh.env.first_env_id == first
    scen_of = lambda gid: 4 + gid % 4                                  # noqa: E731 - the bench's physical workload
    obs = _np(sh.reset(seed=1000, scenario_of=scen_of)).copy()
    pick = np.r_[0:192, n // 2:n // 2 + 128, n - 192:n]                # first, middle, last wavefronts of the shard
    ora = O.OracleBatch(len(pick))
    z = np.stack([O.OracleBatch(1, first_env_id=int(first + i)).normals(1000)[0] for i in pick])
    oobs = ora.reset(ora.mix(means, stds, scen_of(first + pick).astype(np.int32), z))
    assert np.abs(obs[pick] - oobs).max() < 1e-5
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    pool = torch.rand(8, n_global, 2, device="cuda", generator=gen) * torch.tensor([2.5, 15.0], device="cuda")
    big = G.SbrOSVec(n_global)
    gid = torch.arange(n_global, device="cuda")
    big.reset(seed=1000, scenario=scen_of(gid).to(torch.int32))
    double_steps = 0
    for c in range(calls):
        a = pool[c & 7]
        a_sh = a[first:first + n].contiguous()
        o, s_, r, d = sh.step(a_sh)
        big.step(a)
        t_before = ora.envs["t"][0]
        oo, os_, orr, od = ora.step(_np(a_sh)[pick].astype(np.float64))
        double_steps += int(ora.envs["t"][0] - t_before > 1.5 * 0.002 / 24 * 10)
        assert np.array_equal(_np(d)[pick], od)
        if c % 6 == 0 or c >= calls - 2:
            x, _ = sh.env.get_state()
            g = gate(_np(x).T[pick], ora.envs["x"]).max()
            assert g < 1e-6, (c, g)                                    # free-running, every sampled env
            assert np.allclose(_np(r)[pick], orr, rtol=2e-7, atol=1e-10) and np.abs(_np(o)[pick] - oo).max() < 1e-5
    assert double_steps == 1                                           # call 46 ran the last anoxic and the first aerobic interval
    x_sh, c_sh = sh.env.get_state()
    x_big, c_big = big.get_state()
    assert torch.equal(x_sh, x_big[:, first:first + n]) and torch.equal(c_sh, c_big[:, first:first + n])
    assert torch.equal(sh.env.episode_returns(), big.episode_returns()[first:first + n])
    sh.close(); big.close()


def test_indexing_at_134_million_envs(G):
    """Maximum sizes: 2**27 envs in ONE handle (51 GB of plant/controller/influent rows; element indices into the ctrl
    block pass 2**31 from row 16 on, obs offsets pass 2**31 at env 119 M).  Envs are independent and keyed by their global
    id, so three 256-env windows (start, middle, end) must be bit-identical to small handles created with the same
    first_env_id and fed the same actions."""
    from gym_sbr2_amd import _capi
    n, w, calls = 1 << 27, 256, 12
    big = G.SbrOSVec(n, out_dtype=torch.float64)
    scen = (torch.arange(n, device="cuda") % 8).to(torch.int32)
    big.reset(seed=11, scenario=scen)
    bases = [0, (1 << 26) - 128, n - w]
    smalls = []
    for b0 in bases:
        e = G.SbrOSVec(w, first_env_id=b0, out_dtype=torch.float64)
        e.reset(seed=11, scenario=scen[b0:b0 + w].contiguous())
        smalls.append(e)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    for c in range(calls):
        a = torch.rand(n, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")
        o, s, r, d = big.step(a)
        for b0, e in zip(bases, smalls):
            so, ss, sr, sd = e.step(a[b0:b0 + w].contiguous())
            assert torch.equal(o[b0:b0 + w], so) and torch.equal(s[b0:b0 + w], ss), (c, b0)
            assert torch.equal(r[b0:b0 + w], sr) and torch.equal(d[b0:b0 + w], sd), (c, b0)
    for row in (_capi.C_T, _capi.C_KLA_LAST, _capi.C_KLA_HIST0, _capi.C_RETURN, _capi.C_STEPS, _capi.C_QW):
        full = big.ctrl_row(row)
        for b0, e in zip(bases, smalls):
            assert torch.equal(full[b0:b0 + w], e.ctrl_row(row)), (row, b0)
    assert bool((big.ctrl_row(_capi.C_STEPS) == calls).all()) and bool(torch.isfinite(r).all())
    ret = big.episode_returns()
    st = big.stats(ret)                                            # device reduction over all 2**27 values
    assert st["count"] == n and st["min"] == float(ret.min().item()) and st["max"] == float(ret.max().item())
    big.close()
    for e in smalls:
        e.close()


def test_trajectory_export_matches_reference_per_call(G, tables):
    """sbr_set_trace: one record per call for the traced envs.  Compared with what the reference logged per call in the
    six golden episodes (float64 actions), and with the oracle for the traced subset of a bigger batch."""
    from gym_sbr2_amd import _capi
    E, env, ora, acts, _, _, ncall = _run_golden_batch(G, tables, torch.float64)
    tr = env.enable_trace(n_envs=4, capacity=ncall + 5)      # first four episodes only
    env.reset(rnd=np.stack([e["rnd"] for e in E]))            # tracing starts with the next reset
    for c in range(ncall):
        env.step(torch.from_numpy(acts[c]).cuda())
    tr = _np(tr)
    assert np.isnan(tr[ncall:]).all() and np.isfinite(tr[:ncall]).all()
    for i in range(4):
        e = E[i]
        assert np.array_equal(tr[:ncall, 0, i], e["step_t"])                              # time recurrence is exact
        assert np.array_equal(tr[:ncall, 18, i], e["step_done"].astype(float))
        if EPISODES[i] in CLOSED_LOOP_OK:
            assert gate(tr[:ncall - 1, 1:15, i], e["step_x_end"][:ncall - 1]).max() <= 1.0
            assert np.abs(tr[:ncall, 17, i] - e["step_reward"]).max() < 5e-7
            assert np.abs(tr[:ncall - 1, 15, i] - e["step_Kla"][:ncall - 1]).max() < 3e-3      # RK4 vs reference, closed loop: measured <= 3.0e-4 (range 0..240)
    x, ctrl = env.get_state()
    assert np.array_equal(tr[ncall - 1, 1:15, :], _np(x)[:, :4])                          # last record = final state
    env.disable_trace()
    env.close()


def test_multi_cycle_carry_over_against_oracle(G, tables):
    """sbr_reset_carry: the next cycle starts from the state the last one ended in (x0 := x after idle, IV := V).  The
    reference prepares this (x0_new, IV_new) but keeps it disabled, so this is pinned device-vs-oracle only: three
    consecutive cycles of 64 envs with held set-points that keep the plant physical."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 64
    scen = (np.arange(n) % 8).astype(np.int32)
    rs = np.random.RandomState(11)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    ora = O.OracleBatch(n)
    a = np.column_stack([rs.uniform(0.5, 2.5, n), rs.uniform(2, 12, n)])
    vols = []
    for cycle in range(3):
        z = rs.randn(n, 48)
        infl = ora.mix(means, stds, scen, z)
        if cycle == 0:
            obs = _np(env.reset(scenario=scen, rnd=z)); oobs = ora.reset(infl)
        else:
            obs = _np(env.reset(scenario=scen, rnd=z, carry_over=True)); oobs = ora.reset_carry(infl)
        x, ctrl = env.get_state()
        assert np.abs(obs - oobs).max() < 1e-10 and gate(_np(x).T, ora.envs["x"]).max() < 1e-6
        assert np.abs(_np(env.influent()).T - ora.envs["influent"]).max() < 1e-9          # inflow = (WV - IV)/T_fill per env
        assert np.all(_np(ctrl)[_capi.C_DONE] == 0) and np.all(_np(ctrl)[_capi.C_STEPS] == 0)
        for c in range(463):
            o, s, r, d = env.step(torch.from_numpy(a).cuda())
            oo, os_, orr, od = ora.step(a)
        x, ctrl = env.get_state()
        assert np.all(_np(d) == 1) and np.array_equal(_np(d), od)
        clean = ((_np(ctrl)[_capi.C_STATUS].astype(int) | ora.envs["status"].astype(int)) & _capi.ST_NEAR_POLE) == 0
        assert clean.sum() >= n // 2
        assert gate(_np(x).T[clean], ora.envs["x"][clean]).max() < 1e-6
        assert np.abs(_np(ctrl)[_capi.C_RETURN] - ora.envs["ret"])[clean].max() < 1e-10
        vols.append(_np(x)[0].copy())
    assert np.all(vols[0] < 1.32) and not np.allclose(vols[0], vols[1])                     # cycles differ: state is carried
    env.close()


@pytest.mark.parametrize("fixture", ["sbrv2_cycles", "sbrv2_cycles_heldout"])
def test_cycle_env_sbr_v2_against_oracle_and_reference(G, tables, fixture):
    """`SBR-v2`: one step() = one whole 12 h cycle in one launch.  The five reference cycles (float64
    actions incl. an out-of-range one) - and (round 6) eight HELD-OUT ones whose DO set-points sit inside the oxygen knee -, then
    256 envs with random influent/scenarios/actions against the C oracle, then a carried-over second cycle."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    g = golden(fixture)
    n = len(g["actions"])
    env = G.SbrEnv2Vec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    obs0 = _np(env.reset(rnd=g["rnd"])).copy()
    assert np.abs(obs0 - g["reset_state"]).max() < 1e-11                  # sums of start state and influent
    obs, rew, done = env.step(torch.from_numpy(g["actions"]).cuda())
    obs, rew, diag = _np(obs), _np(rew), _np(env.diag)
    x, _ = env.get_state()
    last = g["ph_x_end"][g["phase_first"] + 5]
    assert np.all(_np(done) == 1)
    assert gate(_np(x).T, last).max() <= 0.1                               # oracle: 0.018 (RK4 vs the reference's LSODA)
    assert np.abs(rew - g["reward"]).max() < 1e-6                          # oracle: 3.2e-8
    assert np.allclose(obs, g["state"], rtol=1e-6, atol=1e-7)
    assert np.abs(diag[:, 0] / g["Qw"] - 1).max() < 1e-5 and np.abs(diag[:, 1] / g["EQI"] - 1).max() < 1e-6
    # effluent Ntot, COD, Snh, BOD5, Sno: mixed tolerance of the gate (Sno/Snh decay to ~0 in some cycles, where a purely
    # relative error is meaningless): 1e-5 |ref| + 1e-5 * 20
    assert np.all(np.abs(diag[:, 3:8] - g["eff"][:, 1:]) <= 1e-5 * np.abs(g["eff"][:, 1:]) + 2e-4)
    ora = O.OracleCycleBatch(n)
    from oracle.sbr_ref import influent_mix
    ora.reset(np.stack([influent_mix(means[0], stds[0], g["rnd"][c]) for c in range(n)]))
    ost, orew, odiag = ora.step(g["actions"])
    assert gate(_np(x).T, ora.x).max() < 1e-6 and np.abs(rew - orew).max() < 1e-11 and np.abs(obs - ost).max() < 1e-9
    assert np.allclose(diag, odiag, rtol=1e-10, atol=1e-12)
    env.close()
    # a bigger batch: float32 I/O, all eight scenarios, random actions incl. out-of-range ones
    n = 256
    rs = np.random.RandomState(8)
    scen = (np.arange(n) % 8).astype(np.int32)
    z = rs.randn(n, 48)
    a = rs.uniform(-0.2, 1.2, (n, 3)).astype(np.float32)
    env = G.SbrEnv2Vec(n)
    ora = O.OracleCycleBatch(n, nthreads=8)
    infl = O.OracleBatch(n).mix(means, stds, scen, z)
    o0 = _np(env.reset(scenario=scen, rnd=z)); oo0 = ora.reset(infl)
    assert np.allclose(o0, oo0, rtol=2e-6, atol=1e-6)
    obs, rew, _ = env.step(torch.from_numpy(a).cuda())
    ost, orew, odiag = ora.step(a.astype(np.float64))
    x, ctrl = env.get_state()
    clean = (_np(ctrl)[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) == 0
    assert clean.sum() > n // 2 and gate(_np(x).T[clean], ora.x[clean]).max() < 1e-6
    assert np.allclose(_np(rew)[clean], orew[clean], rtol=2e-6, atol=1e-5) and np.allclose(_np(obs)[clean], ost[clean], rtol=2e-6, atol=1e-4)
    assert np.allclose(_np(env.diag)[clean], odiag[clean], rtol=1e-9, atol=1e-11)
    # second cycle carried over from the end state of the first
    z2 = rs.randn(n, 48)
    env.reset(scenario=scen, rnd=z2, carry_over=True); ora.reset(O.OracleBatch(n).mix(means, stds, scen, z2), carry_over=True)
    obs, rew, _ = env.step(torch.from_numpy(a).cuda()); ost, orew, _ = ora.step(a.astype(np.float64))
    x, ctrl = env.get_state()
    clean &= (_np(ctrl)[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) == 0
    assert clean.sum() > n // 4 and gate(_np(x).T[clean], ora.x[clean]).max() < 1e-6
    assert np.allclose(_np(rew)[clean], orew[clean], rtol=2e-6, atol=1e-5)
    env.close()
    # the reference-shaped single env
    e1 = G.make("SBR-v2")
    s0 = e1.reset(rnd=g["rnd"][0])
    s1, r1, d1, info = e1.step(g["actions"][0])
    assert s0.shape == (3,) and np.abs(s0 - g["reset_state"][0]).max() < 1e-11 and d1 is True and info == {}
    assert abs(r1 - g["reward"][0]) < 1e-6 and np.allclose(s1, g["state"][0], rtol=1e-6) and abs(e1.diagnostics()["Qw"] / g["Qw"][0] - 1) < 1e-5
    e1.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_cycle_env_with_perturbed_constants_and_phase_schedule(G, tables, seed):
    """SBR-v2 with constants drawn within +-20 % of their defaults, the PHASE LENGTHS included: the kernel takes the number of
    control intervals of each phase, linspace's step and the reward's 1/(n td) from the host (derive_params, with the
    reference's own IEEE operations), the oracle forms them per call like the reference does (sub_phases_FB.py:183-184) - they
    must agree for any schedule, not just the default one."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 96
    rs = np.random.RandomState(seed)
    cfg, p = _capi.default_config(), O.default_params()
    for name in PERTURBED + ["cyc_Kc", "cyc_tauI", "cyc_tauD"]:
        v = getattr(cfg, name) * rs.uniform(0.8, 1.2)
        setattr(cfg, name, v); setattr(p, name, v)
    ratios = np.array(list(cfg.t_ratio)) * rs.uniform(0.8, 1.2, 8)
    for k in range(8):
        cfg.t_ratio[k] = p.t_ratio[k] = float(ratios[k])
    scen = (np.arange(n) % 8).astype(np.int32)
    z = rs.randn(n, 48)
    a = rs.uniform(0.0, 0.4, (n, 3))
    env = G.SbrEnv2Vec(n, out_dtype=torch.float64, action_dtype=torch.float64, config=cfg)
    ora = O.OracleCycleBatch(n, params=p, nthreads=8)
    infl = O.OracleBatch(n, params=p).mix(means, stds, scen, z)
    assert np.abs(_np(env.reset(scenario=scen, rnd=z)) - ora.reset(infl)).max() < 1e-9
    obs, rew, _ = env.step(torch.from_numpy(a).cuda())
    ost, orew, odiag = ora.step(a)
    x, ctrl = env.get_state()
    clean = (_np(ctrl)[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) == 0
    g = gate(_np(x).T[clean], ora.x[clean]).max()
    print("per-cycle env, perturbed constants and schedule, seed %d: %d of %d envs clear of the poles, worst gate %.3e" % (seed, clean.sum(), n, g))
    assert clean.sum() > n // 2 and g < 1e-6
    assert np.allclose(_np(rew)[clean], orew[clean], rtol=1e-9, atol=1e-9) and np.allclose(_np(obs)[clean], ost[clean], rtol=1e-9, atol=1e-9)
    # the mean Kla of the three aerated phases and Xf: the interval counts n of the schedule enter here (sum(Kla)/n)
    assert np.allclose(_np(env.diag)[clean], odiag[clean], rtol=1e-9, atol=1e-11)
    env.close()


def _sharded_worker(rank, world, n_global, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)         # one GPU on this box: gloo carries the gather
    try:
        from gym_sbr2_amd.sharding import ShardedSbrOS, gather_returns
        sh = ShardedSbrOS(n_global, rank=rank, world=world, device=0, out_dtype=torch.float32)
        sh.reset(seed=5)
        sh.rollout(120, policy_seed=9)
        local = sh.env.episode_returns().to(torch.float32).cpu()
        full = gather_returns(local, n_global)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
        sh.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_sharded_by_global_env_id(G, tmp_path):
    """BASELINE.json configs[3] rehearsed on one GPU: two processes, each a contiguous shard of the global env ids with
    its own handle; the single collective (all-gather of episode returns) must give every rank the vector a single
    process computes - influent noise, scenarios and the random policy are keyed by the GLOBAL env id."""
    import socket
    import torch.multiprocessing as mp
    n_global, world = 1000, 2                                             # ragged: 500 + 500, not multiples of 64
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sharded_worker, args=(world, n_global, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    env = G.SbrOSVec(n_global)
    env.reset(seed=5, scenario=(np.arange(n_global) % 8).astype(np.int32))
    env.rollout(120, policy_seed=9)
    single = _np(env.episode_returns().to(torch.float32))
    assert np.array_equal(got[0], got[1]) and got[0].shape == (n_global,)
    assert np.array_equal(got[0], single)                                 # bit-identical: same kernel, same per-env inputs
    env.close()


def test_pinned_host_io_equals_device_io(G):
    """step_host(): the kernel reads the action from, and writes obs/state/reward/done to, pinned host memory; same numbers
    as the device-tensor path, bit for bit, over a whole episode including the done call."""
    n = 5
    a_env, b_env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64), \
        G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    views = b_env.enable_host_io()
    assert views[1].shape == (n, 18) and views[2].shape == (n, 15) and b_env._h_obs.is_pinned()
    scen = np.arange(n, dtype=np.int32)
    a_env.reset(seed=3, scenario=scen); b_env.reset(seed=3, scenario=scen)
    rs = np.random.RandomState(0)
    for c in range(463):
        act = np.column_stack([rs.uniform(0, 8, n), rs.uniform(0, 15, n)])
        o, s, r, d = a_env.step(torch.from_numpy(act))
        ho, hs, hr, hd = b_env.step_host(act)
        assert np.array_equal(_np(o), ho) and np.array_equal(_np(s), hs) and np.array_equal(_np(r), hr), c
        assert np.array_equal(_np(d), hd), c
    assert hd.all()
    a_env.close(); b_env.close()


def test_reference_shaped_single_env(G):
    """The N = 1 class keeps the reference's surface: reset() -> (list9, list9); step -> 5-tuple (:438, :1273)."""
    e = golden("sbros_const_2_5")
    env = G.make("SBROS-v1")
    obs = env.reset(rnd=e["rnd"])
    assert isinstance(obs, tuple) and len(obs) == 2 and len(obs[0]) == 9 and len(obs[1]) == 9
    assert np.abs(np.r_[obs[0], obs[1]] - np.r_[e["reset_obs_DO"], e["reset_obs_EC"]]).max() < 1e-6     # RK4 fill phase: measured 2e-8
    total, done, k = 0.0, False, 0
    while not done:
        obs, state, reward, done, info = env.step([2.0, 5.0])
        assert isinstance(state, np.ndarray) and state.shape == (15,) and isinstance(reward, float) and info == {}
        if k in (0, 1, 99):
            assert abs(reward - e["step_reward"][k]) < 1e-5 * abs(e["step_reward"][k])
        total += reward; k += 1
    assert k == 463 and abs(total / float(e["episode_return"]) - 1) < 1e-5      # reference: -0.87896708834557
    # trajectory(): the reference's 18-tuple, in its order (:1288), one entry per call
    (t_t, x_t, u_DO_t, u_EC_t, state_t, So_t, Ss_t, EC, Sno_t, dcv_EC, ie_EC, e_EC, reward_t, reward_EQI_t, reward_OCI_t,
     reward_AE_t, reward_EC_t, Snh_t) = env.trajectory()
    assert x_t.shape == (463, 14) and np.array_equal(t_t, e["step_t"]) and all(
        isinstance(v, list) and len(v) == 463 for v in (t_t, u_DO_t, u_EC_t, state_t, So_t, Ss_t, EC, Sno_t, dcv_EC, ie_EC, e_EC,
                                                        reward_t, reward_EQI_t, reward_OCI_t, reward_AE_t, reward_EC_t, Snh_t))
    assert np.abs(np.array(reward_t) - e["step_reward"]).max() < 5e-7 and abs(sum(reward_t) - total) < 1e-12
    # the four diagnostics module_reward_EQIOCI.py:109-112 appends per call, against the reference's own lists
    assert np.abs(np.array(reward_EQI_t) - e["step_r_EQI2"]).max() < 5e-7 * max(1.0, np.abs(e["step_r_EQI2"]).max())
    assert np.abs(np.array(reward_OCI_t) - e["step_r_OCI2"]).max() < 5e-7
    assert np.abs(np.array(reward_AE_t) - e["step_r_AE2"]).max() < 5e-7
    assert np.abs(np.array(reward_EC_t) - e["step_r_EC2"]).max() < 5e-7
    assert gate(x_t[:462], e["step_x_end"][:462]).max() <= 1.0 and np.array_equal(np.array(So_t), x_t[:, 8])
    # closed loop against the reference's default-tolerance run: the NO3-PID's integral sums (Sno - u_EC) dt, so it carries
    # the gate-level (1e-5) differences of Sno (<= 466 x 3e-4 x dt ~ 1.2e-5; measured 4.2e-8); EC itself is saturated at a
    # clamp on almost every call of this episode and equal there
    assert np.abs(np.array(ie_EC) - e["step_ie_EC"]).max() < 1.2e-5
    assert (np.abs(np.array(EC) - e["step_EC"]) < 1e-9).mean() > 0.95 and np.abs(np.array(EC) - e["step_EC"]).max() <= 5e-4
    # set-points in force: u_EC = clip(action[1]) in the anoxic phases, u_DO = clip(action[0]) in the aerobic ones (:862-906)
    aer = e["iv_kind"][np.searchsorted(e["iv_call"], np.arange(463), side="right") - 1] == 1
    assert np.array_equal(np.array(u_DO_t), np.where(aer, 2.0, 0.0)) and np.array_equal(np.array(u_EC_t), np.where(aer, 0.0, 5.0))
    # e_EC = Sno[-1] - u_EC (:1918, :2006) with Sno[-1] as the previous call left it (calls that run one interval)
    one = e["step_n_intervals"][1:462] == 1
    assert np.abs(np.array(e_EC)[1:462] - (e["step_Sno_m1"][:461] - np.array(u_EC_t)[1:462]))[one].max() < 1e-4
    assert all(np.array_equal(a, b) for a, b in zip(state_t[:3], env._states[:3]))
    d = env.trajectory(as_dict=True)
    assert np.array_equal(d["reward_t"], np.array(reward_t)) and np.abs(d["Kla"] - e["step_Kla"]).max() < 1e-3
    assert [a.tolist() for a in env.get_available_actions([0.05, 12.0], 2, 3)] == [[0.0, 1.0, 1.0], [1.0, 1.0, 0.0]]
    obs2 = env.reset(rnd=e["rnd"], carry_over=True)          # second cycle from where the first one ended
    assert len(obs2[0]) == 9 and obs2[0] != obs[0]
    with pytest.raises(NotImplementedError):
        G.make("SBR-v4")
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scheme", [("const_2_5", 1), ("random_a", 1), ("const_2_5", 0)])
def test_dense_trajectory_on_the_reference_output_grid(G, name, scheme):
    """trajectory(dense=True): the rows the reference appends on its own output grids - 252 over the fill phase (:296-313), per
    control interval t_range[1:], x_out[1:], x_out[:-1, k] and len - 1 copies of the set-points (gym_SBR_oneshot.py:1339,
    :1359-1369, :876-892), constant settle / draw rows and the idle phase on the done call (:1122-1155) - rebuilt from RK4 nodes
    replayed on the device by cubic Hermite interpolation.  Against the reference's OWN lists and LSODA rows of the golden
    episode: the time grid bit for bit (all 4767 entries), the states inside the parity gate (closed loop: an episode the
    default-tolerance run can be followed on)."""
    e = golden("sbros_" + name)
    env = G.make("SBROS-v1", scheme=scheme)
    env.reset(rnd=e["rnd"])
    for k in range(463):
        env.step(e["actions"][k])
    per_call = env.trajectory(as_dict=True)
    d = env.trajectory(as_dict=True, dense=True)
    rows = e["iv_n_rows"]                                   # 9 or 10 output rows per interval, 466 intervals
    n_dense, lo = int((rows - 1).sum()), 252                # the fill phase's 252 rows come first (:296-313)
    n_all = int(e["traj_len_t_t"])
    assert n_dense == 3994 and n_all == 4767 == int(e["traj_len_x_t"])
    # the reference's whole t_t - [0] + 251 fill rows + 8 or 9 per interval + settle, draw and idle rows: the same doubles
    assert len(d["t_t"]) == n_all and np.array_equal(np.array(d["t_t"]), e["traj_t_t"])
    assert d["x_t"].shape == (n_all, 14) and len(d["u_DO_t"]) == n_all and len(d["u_EC_t"]) == n_all
    assert np.array_equal(np.array(d["t_t"][lo:lo + n_dense]), np.concatenate([e["iv_t_rows"][i, 1:rows[i]] for i in range(466)]))
    # control intervals against the reference's LSODA rows
    ref_rows = np.vstack([e["iv_x_rows"][i, 1:rows[i]] for i in range(466)])
    g = gate(d["x_t"][lo:lo + n_dense], ref_rows)
    print("dense rows of %s: worst gate %.3f over %d rows" % (name, g.max(), n_dense))
    assert g.max() <= 1.0, g.max()
    # fill phase: starts at x0_init, ends in the post-fill state; done call: ends in the state after idle
    assert np.array_equal(d["x_t"][0], e["x0_init"]) and gate(d["x_t"][lo - 1], e["x_postfill"]).max() <= 1.0
    assert gate(d["x_t"][-1], e["term_x_after_idle"]).max() <= 1.0
    assert gate(d["x_t"][lo + n_dense + 47], e["term_x_pre_settle"]).max() <= 1.0        # the 48 settle rows hold the pre-settle state
    assert gate(d["x_t"][lo + n_dense + 48], e["term_x_after_draw"]).max() <= 1.0        # the 11 draw rows the drawn reactor
    # the concentration lists take x_out[:-1] (start row in, end row out): the reference's own lists, all 4766 entries
    for key, j, scale in (("So_t", 8, 8.0), ("Sno_t", 9, 20.0), ("Snh_t", 10, 20.0)):
        ref = e["traj_" + key]
        assert len(d[key]) == len(ref) == n_all - 1
        assert (np.abs(np.array(d[key]) - ref) <= 1e-5 * np.abs(ref) + 1e-5 * scale).all(), key
    assert len(d["Ss_t"]) == n_all - 1
    # set-points: len - 1 copies per interval of what was in force in THAT interval (both intervals of a boundary call)
    assert np.array_equal(np.array(d["u_DO_t"][lo:lo + n_dense]), np.repeat(e["iv_u_DO"], rows - 1))
    assert np.array_equal(np.array(d["u_EC_t"][lo:lo + n_dense]), np.repeat(e["iv_u_EC"], rows - 1))
    assert set(d["u_DO_t"][lo + n_dense:]) == {d["u_DO_t"][lo + n_dense - 1]} and not any(d["u_DO_t"][:lo])   # Kla = 0 during the fill
    # the last row of a call's last interval is the state step() returned (the per-call record), to rounding
    ends = lo + np.cumsum(rows - 1)[np.searchsorted(e["iv_call"], np.arange(463), side="right") - 1] - 1
    rel = np.abs(d["x_t"][ends[:462]] - per_call["x_t"][:462]) / (np.abs(per_call["x_t"][:462]) + 1e-9)
    assert rel.max() < 1e-12, rel.max()
    # ... and of the done call, after idle: its rows are replayed from the START of the call's last interval with RK4 nodes
    # (sbr_eval_substeps), while step() integrated that interval with the handle's scheme; under scheme 1 the two end states
    # differ by the two discretisations' distance (~1e-2 of the gate: 3e-7 .. 3e-6 relative after the idle phase; 1e-9 with
    # cfg.scheme = 0, where the replay repeats the step's own arithmetic)
    rel = np.abs(d["x_t"][-1] - per_call["x_t"][462]) / (np.abs(per_call["x_t"][462]) + 1e-9)
    # the 1e-9 of the RK4 replay stays pinned where the replay repeats the step's own arithmetic (cfg.scheme = 0)
    assert rel.max() < (2e-5 if scheme == 1 else 1e-9) and gate(d["x_t"][-1], per_call["x_t"][462]).max() < 0.1, rel.max()
    # a running episode: dense rows exist from the first call on
    env2 = G.make("SBROS-v1")
    env2.reset(rnd=e["rnd"])
    assert len(env2.trajectory(as_dict=True, dense=True)["t_t"]) == 252
    env2.step(e["actions"][0])
    assert len(env2.trajectory(as_dict=True, dense=True)["t_t"]) == 252 + int(rows[0]) - 1
    env2.close()
    # calls after `done` are ignored by a finished env and leave no rows; a carried-over second cycle fills from where the first
    # one ended (its dense rows start at that state and pass through the new post-fill state)
    env.step(e["actions"][0])
    assert len(env.trajectory(as_dict=True, dense=True)["t_t"]) == n_all
    x_end = per_call["x_t"][462].copy()                  # the state the done call left on the device
    env.reset(rnd=e["rnd"], carry_over=True)
    for k in range(3):
        env.step(e["actions"][k])
    d2 = env.trajectory(as_dict=True, dense=True)
    assert np.array_equal(d2["x_t"][0], x_end)
    assert np.allclose(d2["x_t"][251], env._x_postfill, rtol=1e-9, atol=1e-12)       # the replayed fill ends where the device's did
    assert abs(d2["x_t"][251][0] - 1.32) < 1e-12 and len(d2["t_t"]) == 252 + int((rows[:3] - 1).sum())
    assert np.abs(d2["x_t"][-1] - env.trajectory(as_dict=True)["x_t"][2]).max() < 1e-9
    # the lists that are per call upstream too are untouched by dense=True
    assert np.array_equal(d["reward_t"], per_call["reward_t"]) and len(per_call["EC"]) == 463
    # ---- the controller lists as the reference grows them (gym_SBR_oneshot.py:1918-1957, :2006-2045, :320-324)
    # EC: [0, 0] * 126 over the fill phase, len(t_range) - 1 copies of every INTERVAL's EC (both intervals of a boundary
    # call), zeros over settle / draw / idle: the reference's list entry for entry
    ref_EC = e["traj_EC"]
    assert len(d["EC"]) == len(ref_EC) == n_all
    dEC = np.array(d["EC"], dtype=np.float64)
    assert np.array_equal(dEC[:lo], ref_EC[:lo]) and np.array_equal(dEC[lo + n_dense:], ref_EC[lo + n_dense:]) and not dEC[:lo].any()
    diff = np.abs(dEC[lo:lo + n_dense] - ref_EC[lo:lo + n_dense])
    assert (diff < 1e-9).mean() > 0.95 and diff.max() <= 5e-4           # saturated at a clamp on almost every interval, equal there
    assert np.array_equal(dEC[lo:lo + n_dense], np.repeat(dEC[lo + np.concatenate([[0], np.cumsum(rows - 1)[:-1]])], rows - 1))
    # e_EC, ie_EC, dcv_EC: the fill phase's entry, then ONE per interval = 467 entries (463 calls, 3 of them with two intervals)
    for key in ("e_EC", "ie_EC", "dcv_EC"):
        assert len(d[key]) == len(e["traj_" + key]) == 467, key
    assert d["e_EC"][0] == e["traj_e_EC"][0] == -e["x0_init"][9] and d["ie_EC"][0] == 0.0 and d["dcv_EC"][0] == 0.0
    # Sno[-1] of interval i is the reference's own to the parity gate (closed loop), e = Sno[-1] - u_EC, dcv = (Sno[-1] - Sno[-2])/dt
    tol_sno = 1e-5 * np.abs(e["traj_e_EC"][1:] + e["iv_u_EC"]) + 1e-5 * 20.0
    assert (np.abs(np.array(d["e_EC"][1:]) - e["traj_e_EC"][1:]) <= tol_sno).all()
    assert np.abs(np.array(d["ie_EC"]) - e["traj_ie_EC"]).max() < 1.2e-5           # the integral carries the gate-level differences
    dt_ = 0.002 / 24
    tol_dcv = (tol_sno + np.concatenate([[tol_sno[0]], tol_sno[:-1]])) / dt_
    assert (np.abs(np.array(d["dcv_EC"][1:]) - e["traj_dcv_EC"][1:]) <= tol_dcv).all()
    assert abs(d["dcv_EC"][1] - e["traj_dcv_EC"][1]) < 1e-6 * abs(e["traj_dcv_EC"][1])   # (Ss after the fill - x0[9]) / dt: the :1652 quirk
    # the boundary calls contribute two entries: entry k of the lists belongs to interval k - 1 = (call iv_call[k - 1])
    two = np.nonzero(e["step_n_intervals"] == 2)[0]
    assert len(two) == 3
    for c in two:
        i1 = int(np.nonzero(e["iv_call"] == c)[0][0])         # first interval of that call
        assert d["e_EC"][1 + i1] != d["e_EC"][2 + i1] and per_call["e_EC"][c] == d["e_EC"][2 + i1]    # per-call list = the last interval's
    env.close()


@pytest.mark.gpu
def test_random_scenario_is_drawn_on_the_device(G, tables):
    """cfg.random_scenario = 1: reset(scenario=None) gives every env one of the 8 influent scenarios, uniformly, as
    SbrEnv4.reset does with np.random.choice(8, 1) (gym_SBR_env4.py:107) - Philox stream 2 keyed by the seed of the reset
    and the GLOBAL env id, so a shard draws what the whole batch would."""
    means, stds = tables
    n = 4096
    env = G.SbrOSVec(n, out_dtype=torch.float64, random_scenario=True)
    ora = O.OracleBatch(n)
    for seed in (0, 17):
        sc = _np(env.draw_scenarios(seed))
        assert np.array_equal(sc, ora.scenarios(seed)) and sc.min() == 0 and sc.max() == 7
        assert np.abs(np.bincount(sc, minlength=8) / n - 0.125).max() < 0.03          # uniform: 4 sigma is 0.021
        env.reset(seed=seed)                                                          # scenario=None: drawn on the device
        want = ora.mix(means, stds, sc, ora.normals(seed))
        assert np.abs(_np(env.influent()).T[:, 1:] - want[:, 1:]).max() < 1e-11
    assert not np.array_equal(_np(env.draw_scenarios(0)), _np(env.draw_scenarios(17)))
    win = G.SbrOSVec(100, first_env_id=1000, out_dtype=torch.float64, random_scenario=True)    # a shard's window
    assert np.array_equal(_np(win.draw_scenarios(17)), _np(env.draw_scenarios(17))[1000:1100])
    win.reset(seed=17)
    assert np.array_equal(_np(win.influent()), _np(env.influent())[:, 1000:1100])
    fixed = G.SbrOSVec(64, out_dtype=torch.float64)                                   # default: scenario 6, as SbrOS (:180)
    fixed.reset(seed=3)
    assert np.abs(_np(fixed.influent()).T[:, 1:] - O.OracleBatch(64).mix(means, stds, [6] * 64, O.OracleBatch(64).normals(3))[:, 1:]).max() < 1e-11
    env.close(); win.close(); fixed.close()


@pytest.mark.gpu
def test_results_do_not_depend_on_wave_mates_or_shard_boundaries(G, tables):
    """An env's arithmetic is independent of which envs share its wavefront: the same global envs stepped (a) as one batch,
    (b) as shards cut at a non-multiple of 64, (c) alone, give BIT-IDENTICAL float64 plant and controller state through the
    anoxic phase (where some lanes dose carbon and others do not, i.e. both integrator code paths occur inside a wave),
    a phase boundary and the aerobic phase.  (The first version folded 1/V into the batched reciprocal of the dosing path
    only, so a non-dosing lane rounded differently next to a dosing wave-mate.)"""
    n, cut, calls = 200, 77, 70
    rs = np.random.RandomState(5)
    # NO3 set-point 0 => error Sno - 0 > 0 => the valve opens; set-point 15 => error < 0 => EC stays clamped at 0
    acts = np.stack([np.column_stack([rs.uniform(0, 8, n), np.where(rs.rand(n) < 0.5, 0.0, 15.0)]) for _ in range(calls)])
    scen = (np.arange(n) % 8).astype(np.int32)

    def run(lo, hi):
        env = G.SbrOSVec(hi - lo, first_env_id=lo, out_dtype=torch.float64, action_dtype=torch.float64)
        env.reset(seed=9, scenario=scen[lo:hi])
        ec_mid = None
        for c in range(calls):
            env.step(torch.from_numpy(acts[c, lo:hi]).cuda())
            if c == 30:                                        # inside the first anoxic phase
                ec_mid = _np(env.ctrl_row(7)).copy()
        x, ctrl = env.get_state()
        out = (_np(x).copy(), _np(ctrl).copy(), ec_mid)
        env.close()
        return out
    xa, ca, eca = run(0, n)
    for w in range(0, n - 63, 64):                             # EC[-1] on an anoxic call: every full wave mixes dosing and idle lanes
        assert (eca[w:w + 64] == 0).any() and (eca[w:w + 64] != 0).any(), w
    assert np.all(ca[0] > 0.0641667)                           # and the run went on into the aerobic phase (t > T3_0)
    for lo, hi in ((0, cut), (cut, n), (130, 131)):
        xs, cs, ecs = run(lo, hi)
        assert np.array_equal(xs, xa[:, lo:hi]) and np.array_equal(cs, ca[:, lo:hi]) and np.array_equal(ecs, eca[lo:hi]), (lo, hi)


@pytest.mark.gpu
@pytest.mark.parametrize("reward", ["eqi_oci", "oci"])
def test_parked_build_of_k_step_is_bit_identical_over_a_whole_episode(G, reward):
    """Above 65 536 envs the scheme-1 k_step runs as k_step<..., WAVES = 2>: what a call keeps across the step loops is parked
    in LDS (13 controller values per interval, 7 of the call, 19 around the idle phase of the done call) so that two waves fit a
    SIMD.  Same arithmetic: a 65 536 + 320-env handle (ragged last workgroup) and handles of 4 096 envs (64-thread workgroups)
    / 65 536 envs (256-thread workgroups, one wave per SIMD) on the same global env ids give the same outputs at sampled
    calls, at the phase-boundary calls (two intervals) and at the done call (terminal phases), and the same plant, controller
    rows and returns after the episode - bit for bit, under the reference's reward and the OCI reward (whose running sum takes
    part in the parking)."""
    n_big, calls = 65536 + 320, 463
    gen = torch.Generator(device="cuda"); gen.manual_seed(31)
    pool = torch.rand(16, n_big, 2, device="cuda", generator=gen) * torch.tensor([8.0, 15.0], device="cuda")
    pool[:, ::3, 1] = 0.0                                              # a third of the lanes doses carbon in the anoxic phases
    scen = (torch.arange(n_big, device="cuda") % 8).to(torch.int32)
    views = {"small": (n_big - 4096, n_big), "one_wave": (0, 65536)}
    big = G.SbrOSVec(n_big, reward=reward)
    envs = {k: G.SbrOSVec(hi - lo, first_env_id=lo, reward=reward) for k, (lo, hi) in views.items()}
    # the trajectory export (which also switches the kernels to the So[-2] / Sno[-2] rows) of the first 96 envs, on both builds
    tr_big, tr_one = big.enable_trace(96, calls), envs["one_wave"].enable_trace(96, calls)
    ob = big.reset(seed=17, scenario=scen)
    for k, (lo, hi) in views.items():
        assert torch.equal(envs[k].reset(seed=17, scenario=scen[lo:hi].contiguous()), ob[lo:hi])
    n_done = 0
    for c in range(calls):
        a = pool[c & 15]
        o, s_, r, d = big.step(a)
        check = c % 29 == 0 or c in (45, 46, 47, 276, 277, 278, 279) or c >= calls - 2
        for k, (lo, hi) in views.items():
            o2, s2, r2, d2 = envs[k].step(a[lo:hi].contiguous())
            if check:
                assert torch.equal(o2, o[lo:hi]) and torch.equal(s2, s_[lo:hi]) and torch.equal(r2, r[lo:hi]) and torch.equal(d2, d[lo:hi]), (k, c)
        n_done += int(d.all())
    assert n_done == 1 and bool(d.all())                               # the last call was the done call of every env
    assert torch.equal(torch.nan_to_num(tr_big, nan=-7.0), torch.nan_to_num(tr_one, nan=-7.0)) and not torch.isnan(tr_big[:, 0]).any()
    xb, cb = big.get_state()
    for k, (lo, hi) in views.items():
        x2, c2 = envs[k].get_state()
        assert torch.equal(x2, xb[:, lo:hi]) and torch.equal(c2, cb[:, lo:hi]), k
        assert torch.equal(envs[k].episode_returns(), big.episode_returns()[lo:hi]), k
        envs[k].close()
    big.close()


@pytest.mark.gpu
@pytest.mark.parametrize("substeps", [5, 20])
def test_other_substep_counts_against_oracle(G, tables, substeps):
    """cfg.substeps is a parameter of both the kernels and the oracle (the step h = span / substeps reaches the device as a
    host-side reciprocal): 90 calls across the anoxic -> aerobic boundary for 5 and 20 substeps per interval, lockstep."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n, calls = 192, 90
    scen = (np.arange(n) % 8).astype(np.int32)
    cfg = _capi.default_config(); cfg.substeps = substeps; cfg.scheme = 0
    p = O.default_params(scheme=0); p.substeps = substeps
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64, config=cfg)
    ora = O.OracleBatch(n, params=p)
    rnd = np.random.RandomState(substeps).randn(n, 48)
    obs = _np(env.reset(scenario=scen, rnd=rnd)).copy()
    assert np.abs(obs - ora.reset(ora.mix(means, stds, scen, rnd))).max() < 1e-10
    rs = np.random.RandomState(3)
    for c in range(calls):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)])
        x, ctrl = env.get_state()
        ora.load_state(_np(x), _np(ctrl))
        o, s_, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = ora.step(a)
        x, ctrl = env.get_state()
        assert gate(_np(x).T, ora.envs["x"]).max() < 1e-6 and np.array_equal(_np(d), od), c
        assert np.abs(_np(r) - orr).max() < 1e-12 and np.abs(_np(o) - oo).max() < 1e-10
        assert np.array_equal(_np(ctrl)[_capi.C_T], ora.envs["t"])
    assert float(_np(ctrl)[_capi.C_T].min()) > 0.0641667            # the run crossed into the aerobic phase
    env.close()


@pytest.mark.gpu
def test_c_abi_from_plain_c_without_python_or_torch(G, tmp_path):
    """The drop-in boundary is a C ABI: examples/c_abi_demo.c (C99, gcc, the HIP runtime's C API for device memory) runs the
    reference's own default episode - scenario 6, numpy seed 0, action [2.0, 5.0] - through libsbr_amd.so in a process that
    contains neither Python nor torch, and prints the numbers SURVEY.md 8c lists as anchors.  Compared with the reference's."""
    import shutil, subprocess
    from conftest import ROOT
    from gym_sbr2_amd import build as B
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.dirname(B.LIB)
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", libdir, "-lsbr_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    out = dict(line.split() for line in p.stdout.strip().splitlines())
    e = golden("sbros_const_2_5")
    assert int(out["calls"]) == 463
    for k, call in (("reward[1]", 0), ("reward[2]", 1), ("reward[100]", 99)):
        assert abs(float(out[k]) - e["step_reward"][call]) < 5e-7, k              # -0.005341505578026706, 0.0005042630758199245 upstream
    assert abs(float(out["Kla[100]"]) - e["step_Kla"][99]) < 3e-3                 # 160.969686931186
    assert abs(float(out["So[100]"]) - e["step_x_end"][99][8]) < 1e-5 * (abs(e["step_x_end"][99][8]) + 8.0)   # 2.0044944036975214
    assert abs(float(out["return"]) / float(e["episode_return"]) - 1) < 1e-5      # -0.8789670883455737
    assert abs(float(out["return_row"]) - float(out["return"])) < 1e-12
    assert abs(float(out["Qw"]) / float(e["term_Qw"]) - 1) < 1e-5                 # 0.05015591126665638


PERTURBED = ("Ya Yh fp ixb ixp muH Ks Koh Kno bH eta_g eta_h kh Kx muA Knh bA Koa ka So_sat Kla_max Kc_DO tauI_DO EC_max Kc_EC tauI_EC "
             "EC_conc act_DO_max act_EC_max biomass_setpoint settler_vmax").split()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_perturbed_model_constants_and_derivative_action_against_oracle(G, tables, seed):
    """Every constant of the model is a parameter of the C ABI, and the kernels do not use them as the reference writes them:
    mu_H, mu_A, k_h and eta_g K_OH are folded into the Monod denominators, b_H and 1/Y_A into step coefficients, the
    stoichiometric products are formed on the host (sbr_rates, derive_params).  That algebra must hold for ANY constants, not
    just the defaults all other tests run with: 31 kinetic, stoichiometric, controller and settler constants are drawn within
    +-20 % of their defaults, and both PIDs get a non-zero derivative time - tau_D = 0 upstream (:85, :94), so the So[-2] /
    Sno[-2] rows and the derivative terms (:1892, :1898-1902, :2016-2025) are otherwise never exercised.  A whole episode in
    lockstep with the oracle (re-synchronised to the device state before every call), terminal phases included."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 128
    rs = np.random.RandomState(seed)
    cfg, p = _capi.default_config(), O.default_params()
    for name in PERTURBED:
        v = getattr(cfg, name) * rs.uniform(0.8, 1.2)
        setattr(cfg, name, v); setattr(p, name, v)
    for name, v in (("tauD_DO", 2e-5 * rs.uniform(0.5, 1.5)), ("tauD_EC", 1e-8 * rs.uniform(0.5, 1.5))):
        setattr(cfg, name, v); setattr(p, name, v)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64, config=cfg)
    ora = O.OracleBatch(n, params=p)
    scen = (np.arange(n) % 8).astype(np.int32)
    rnd = rs.randn(n, 48)
    obs = _np(env.reset(scenario=scen, rnd=rnd)).copy()
    assert np.abs(obs - ora.reset(ora.mix(means, stds, scen, rnd))).max() < 1e-10
    x, ctrl = env.get_state()
    assert gate(_np(x).T, ora.envs["x"]).max() < 1e-6                    # the fill phase under the perturbed constants
    worst, worst_near, d_term = 0.0, 0.0, 0
    for c in range(463):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)])
        x, ctrl = env.get_state()
        ora.load_state(_np(x), _np(ctrl))
        o, s_, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = ora.step(a)
        x, ctrl = env.get_state()
        ctrl = _np(ctrl)
        gi = gate(_np(x).T, ora.envs["x"]).max(axis=1)
        near = (ctrl[_capi.C_STATUS].astype(int) & _capi.ST_NEAR_POLE) != 0      # perturbed kinetics push some envs towards a pole
        worst, worst_near = max(worst, gi[~near].max(initial=0.0)), max(worst_near, gi[near].max(initial=0.0))
        assert gi[~near].max(initial=0.0) < 1e-6 and gi.max() < 1e-4 and np.array_equal(_np(d), od), (c, gi.max())
        assert np.abs(_np(r) - orr).max() < 1e-11 and np.abs(_np(o) - oo).max() < 1e-9 and np.abs(_np(s_) - os_).max() < 1e-9
        assert np.array_equal(ctrl[_capi.C_T], ora.envs["t"])
        # both controllers' memories, derivative inputs included
        for row, key in ((_capi.C_SO_M1, "so_m1"), (_capi.C_SO_M2, "so_m2"), (_capi.C_SNO_M1, "sno_m1"), (_capi.C_SNO_M2, "sno_m2"),
                         (_capi.C_IE_DO, "ie_do"), (_capi.C_IE_EC, "ie_ec"), (_capi.C_EC_LAST, "ec_last"), (_capi.C_KLA_LAST, "kla_last")):
            assert np.allclose(ctrl[row], ora.envs[key], rtol=1e-9, atol=1e-12), (c, key)
        d_term += int(od.all())
    assert d_term == 1 and np.all(ctrl[_capi.C_DONE] == 1) and np.abs(ctrl[_capi.C_QW] / ora.envs["qw"] - 1).max() < 1e-9
    print("perturbed constants, seed %d: lockstep worst gate %.3e (envs near a pole: %.3e)" % (seed, worst, worst_near))
    env.close()


@pytest.mark.gpu
def test_trace_of_several_envs_and_cycle_env_scenario_draw(G, tables):
    """Trajectory export for more than one env of a ragged batch (records of the first n_trace envs only, per-env columns),
    and cfg.random_scenario in the per-cycle env (sbr_cycle_reset draws like sbr_reset)."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n, nt, calls = 100, 37, 56
    scen = (np.arange(n) % 8).astype(np.int32)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    tr = env.enable_trace(n_envs=nt, capacity=calls)
    env.reset(seed=2, scenario=scen)
    ora = O.OracleBatch(n)
    ora.reset(ora.mix(means, stds, scen, ora.normals(2)))
    rs = np.random.RandomState(8)
    rew, plans = [], []
    for c in range(calls):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)])
        o, s_, r, d = env.step(torch.from_numpy(a).cuda())
        ora.step(a)
        rew.append(_np(r).copy()); plans.append(ora.envs["scheme_plan"].copy())
    x, ctrl = env.get_state()
    t = _np(tr)
    assert t.shape == (calls, _capi.NTRACE, nt) and np.isfinite(t).all()
    # round 6: what cfg.scheme = 1 did, per call, for the call's last and first interval (call 51 crosses the anoxic -> aerobic
    # boundary and runs two: slaved two-step interval first, then the aeration switch-on in the knee) - equal to the oracle's plan
    plans = np.array(plans)[:, :nt]
    assert np.array_equal(t[:, _capi.TR_PLAN, :], plans & 0xff) and np.array_equal(t[:, _capi.TR_PLAN_FIRST, :], plans >> 8)
    assert np.all(t[51, _capi.TR_N_IV, :] == 2) and np.all((t[51, _capi.TR_PLAN_FIRST, :].astype(int) & _capi.PLAN_SLAVED) != 0) and np.all((t[51, _capi.TR_PLAN, :] >= 4) & (t[51, _capi.TR_PLAN, :] < 64))
    assert np.array_equal(t[:51, _capi.TR_PLAN, :], t[:51, _capi.TR_PLAN_FIRST, :]) and np.array_equal(_np(ctrl)[_capi.C_PLAN, :nt], t[-1, _capi.TR_PLAN, :])
    assert np.array_equal(t[:, _capi.TR_REWARD, :], np.array(rew)[:, :nt])
    assert np.array_equal(t[-1, _capi.TR_X0:_capi.TR_X0 + 14, :], _np(x)[:, :nt]) and np.array_equal(t[-1, _capi.TR_T, :], _np(ctrl)[_capi.C_T, :nt])
    assert np.allclose(t[:, _capi.TR_R_OCI, :], t[:, _capi.TR_R_AE, :] + t[:, _capi.TR_R_EC, :], rtol=0, atol=1e-15)
    env.disable_trace(); env.close()
    cyc = G.SbrEnv2Vec(256, out_dtype=torch.float64, random_scenario=True)
    cyc.reset(seed=21)
    sc = O.OracleBatch(256).scenarios(21)
    want = O.OracleBatch(256).mix(means, stds, sc, O.OracleBatch(256).normals(21))
    assert len(set(sc.tolist())) == 8 and np.abs(_np(cyc.influent()).T[:, 1:] - want[:, 1:]).max() < 1e-11
    cyc.close()


@pytest.mark.gpu
def test_waves_with_envs_at_different_points_of_their_episodes(G, tables):
    """Masked resets put the lanes of one wavefront at different points of their episodes: different phases (some lanes dose
    carbon while others aerate), different ring positions of the Kla history (the per-lane addressing path of k_step), phase
    boundaries crossed on different calls, episodes ending on different calls.  300 envs, three staggered groups, every call
    followed by the oracle free-running (reset with the same influent when the device group is reset)."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 300
    scen = (4 + np.arange(n) % 4).astype(np.int32)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    ora = O.OracleBatch(n)
    rnd = np.random.RandomState(11).randn(n, 48)
    infl = ora.mix(means, stds, scen, rnd)
    env.reset(scenario=scen, rnd=rnd)
    ora.reset(infl)
    group = np.arange(n) % 3
    rs = np.random.RandomState(12)

    def step_both(c):
        a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)])
        o, s_, r, d = env.step(torch.from_numpy(a).cuda())
        oo, os_, orr, od = ora.step(a)
        assert np.array_equal(_np(d), od), c
        assert np.abs(_np(r) - orr).max() < 1e-11 and np.abs(_np(o) - oo).max() < 1e-9, c
        return _np(d)

    def reset_group(g):
        m = (group == g)
        env.reset(scenario=scen, rnd=rnd, mask=m.astype(np.uint8))
        sub = O.OracleBatch(int(m.sum()))
        sub.reset(infl[m])
        ora.envs[m] = sub.envs

    for c in range(40):
        step_both(c)
    reset_group(1)                     # group 1 restarts 40 calls behind
    for c in range(40, 130):
        step_both(c)
    reset_group(2)                     # group 2 restarts 130 calls behind: anoxic next to the others' aerobic phase
    done_seen = np.zeros(n, bool)
    for c in range(130, 520):          # group 0 ends on call 463, group 1 on 40 + 463 = 503, group 2 would on 593
        done_seen |= step_both(c).astype(bool)
        if c in (200, 300, 400, 470, 510):
            x, ctrl = env.get_state()
            assert gate(_np(x).T, ora.envs["x"]).max() < 1e-6, c
            assert np.array_equal(_np(ctrl)[_capi.C_T], ora.envs["t"]) and np.array_equal(_np(ctrl)[_capi.C_STEPS], ora.envs["steps"])
            assert np.abs(_np(ctrl)[_capi.C_KLA_HIST0:_capi.C_KLA_HIST0 + 10].T - ora.envs["kla_hist"]).max() < 1e-9
    assert done_seen[group == 0].all() and done_seen[group == 1].all() and not done_seen[group == 2].any()
    x, ctrl = env.get_state()
    assert np.abs(_np(ctrl)[_capi.C_RETURN] - ora.envs["ret"]).max() < 1e-10
    env.close()


@pytest.mark.gpu
def test_output_rows_to_misaligned_destinations(G):
    """Full wavefronts write their 64 observation / state rows as one contiguous block with 16-byte stores when the
    destination is 16-byte aligned; any other alignment (a caller's tensor view) takes the per-element form.  Same values."""
    import ctypes as C
    n = 256
    scen = (np.arange(n) % 8).astype(np.int32)
    a = torch.rand(n, 2, device="cuda") * torch.tensor([2.5, 15.0], device="cuda")
    outs = []
    for shift in (0, 1, 2, 3):                       # floats of offset: 0 = aligned, 1..3 = 4, 8, 12 bytes off
        env = G.SbrOSVec(n)
        env.reset(seed=4, scenario=scen)
        obs = torch.full((n * 18 + 8,), -7.0, device="cuda"); st = torch.full((n * 15 + 8,), -7.0, device="cuda")
        for _ in range(3):
            rc = env.lib.sbr_step(env._h, C.c_void_p(a.data_ptr()), C.c_void_p(obs.data_ptr() + 4 * shift),
                                  C.c_void_p(st.data_ptr() + 4 * shift), C.c_void_p(env.reward.data_ptr()),
                                  C.c_void_p(env.done.data_ptr()), None)
            assert rc == 0
        torch.cuda.synchronize()
        o, s_ = _np(obs), _np(st)
        assert (o[:shift] == -7).all() and (o[shift + n * 18:] == -7).all() and (s_[:shift] == -7).all() and (s_[shift + n * 15:] == -7).all()
        outs.append((o[shift:shift + n * 18].copy(), s_[shift:shift + n * 15].copy()))
        env.close()
    for o, s_ in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(s_, outs[0][1])
    assert np.isfinite(outs[0][0]).all() and not (outs[0][0] == -7).any()


@pytest.mark.gpu
@pytest.mark.parametrize("force_dist", [False, True, "launcher"])
def test_bench_line_contract(force_dist):
    """bench.py as the driver runs it (`--gpus 1 --steps 20 --warmup 5`), in its own process; with force_dist the process group,
    the barrier of the bracket and the all-gather run over RCCL with one rank (what every rank of an N > 1 run executes);
    "launcher": the same under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1`, the
    driver's command line for N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment).
    Checks the ONE JSON line: the contract's keys, value = envs * K / wall, the event-timed launch inside the wall time."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, SBR_BENCH_FORCE_DIST="1" if force_dist else "0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    if force_dist == "launcher":
        args = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29541"] + args
        env.pop("MASTER_PORT")
    p = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    t = r["timed_region_ms"]
    assert r["launches_timed"] == 20 and abs(t["step_kernels_device"] - 20 * r["avg_launch_us"] * 1e-3) < 1e-9
    assert t["step_kernels_device"] <= t["wall"] and t["host_issue"] <= t["wall"]
    assert 8.0 < r["avg_launch_us"] < 40.0 and d["value"] > 1e9            # sanity: the order of magnitude of this kernel
    # the one collective of the path (configs[3]): a 20-step region holds no episode boundary, so an N > 1 run issues it once
    # after the K-th step, inside the timed region; an N = 1 run without a process group has none
    c = d["config"]
    assert c["resets_in_timed_region"] == 0
    if force_dist:
        assert c["allgathers_in_timed_region"] >= 1 and c["allgather_bytes_per_rank"] == 4 * 65536
        assert c["ranks"] == 1 and c["backend_reported"] == "nccl" and len(c["rank_elapsed_ms"]) == 1 and c["rank_skew_ms"] == 0.0
    else:
        assert c["allgathers_in_timed_region"] == 0 and c["allgather_bytes_per_rank"] == 0
        assert c["ranks"] == 1 and c["backend_reported"] is None and c["rank_devices"][0].startswith("cuda:0 ")
    # `frac` is the CONSERVATIVE figure - the whole-episode trace of this very library when one is committed
    # (`frac_episode`), never above what this run's wall clock allows - and the flattering one travels beside it
    wall = r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 8e12
    assert abs(r["frac_wall"] - wall) < 1e-9 and r["frac"] <= wall * 1.05 and r["frac"] <= r["frac_timed_launches"] * 1.02
    assert r["frac_is"] in ("frac_episode", "frac_wall") and abs(r["frac"] - r[r["frac_is"]]) < 1e-12
    assert ("frac_episode" in r) and (r["frac_episode"] is None or 0.1 < r["frac_episode"] < 0.6) and r["frac_episode_source"]
    assert c["scheme"] == 1 and c["step_issue"].startswith("HIP-graph") and c["kernel"] == "k_step<float,float,256,false,1,1>"
    # the driver's 20 timed calls straddle the first anoxic -> aerobic boundary with the episode's own mix -
    # ten anoxic calls, the double-step call 51, nine aerobic ones (until round 5: calls 5..24, all anoxic)
    assert c["timed_calls"] == [41, 61] and c["anoxic_share_of_timed_calls"] == 0.5 and abs(c["anoxic_share_of_an_episode"] - 238 / 463) < 1e-12
    # what scheme 1 did on the timed envs is counted on the DEVICE (the plan row), not by the CPU oracle
    b5 = r["fp64_valu"]["b5_steps_per_interval"]
    assert b5["source"].startswith("device") and 1.0 <= b5["per_env_mean"] <= b5["per_wavefront_mean"] <= 8.0 and 0.3 < b5["slaved_share"] < 0.6
    assert 0.0 <= c["dosing_wave_call_share"] <= 1.0
    # ... and so does the no-overlap bound they are to be read against (memory at the roofline's rate + arithmetic + launch floor),
    # from a committed record of this library (None when there is none)
    sb = r["serial_bound"]
    assert sb is None or (abs(sb["sum_us"] - (sb["memory_us"] + sb["arithmetic_us_episode_mean"] + sb["dependent_launch_floor_us"])) < 1e-9
                          and 0.25 < sb["frac_at_bound"] < 0.6 and abs(sb["memory_us"] - r["traffic"] / 8e6) < 1e-6)
    # north_star's ">= 40 % of HBM roofline on one MI355X", measured INSIDE this command on a second handle of
    # 262144 envs (k_step's two-waves-per-SIMD build); committed records of other sizes travel beside it, labelled as such
    lb = r["larger_batches"]
    if force_dist:
        assert lb is None or "262144" not in lb or not lb["262144"].get("measured_in_this_run")
    else:
        big = lb["262144"]
        assert big["measured_in_this_run"] is True and big["envs_per_launch"] == 262144 and big["launches_timed"] == 20
        assert big["kernel"] == "k_step<float,float,256,false,1,2>" and big["timed_calls"] == [41, 61]
        assert abs(big["frac_wall"] - 513 * 262144 / (big["ms_per_step"] * 1e-3) / 8e12) < 1e-9 and big["frac"] == big["frac_wall"]
        assert abs(big["env_steps_per_s"] - 262144 / (big["ms_per_step"] * 1e-3)) < 1e-6 * big["env_steps_per_s"]
        assert big["frac_wall"] >= 0.40, big                       # the north-star fraction, on the clock of this very run
        assert big["frac_timed_launches"] >= big["frac_wall"] * 0.98 and 20.0 < big["avg_launch_us"] < 60.0
        assert 1.0 <= big["b5_steps_per_interval"]["per_env_mean"] <= big["b5_steps_per_interval"]["per_wavefront_mean"] <= 8.0
        for k_, v in lb.items():
            if k_ != "262144":
                assert v["measured_in_this_run"] is False and v["committed_constant"] is True and v["file"].startswith("profiles/") and 0.1 < v["frac"] < 0.8
    # PMC traffic is a committed constant: present only if profiles/ holds a profile of THIS library (same source hash)
    assert (r["traffic"] is None) or ("committed constant" in r["traffic_unit"] and r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"])
    assert r["traffic"] is not None or r["traffic_unit"]


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["walk", "uniform"])
def test_bench_secondary_policy_lines(policy):
    """`--policy walk` (the reference's own action model, get_available_actions :440-459) and `--policy uniform`
    (SURVEY.md 8d's synthetic inputs) are SECONDARY bench lines: same contract, labelled, never the headline."""
    import json, os, subprocess, sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--policy", policy, "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                        "--no-large-leg"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c, r = d["config"], d["roofline"]
    assert c["policy"] == policy and c["timed_calls"] == [36, 66] and d["steps"] == 30
    assert r["larger_batches"] is None                  # neither the in-run leg (--no-large-leg) nor committed lines of the default workload
    assert ("get_available_actions" in c["actions"]) == (policy == "walk") and ("0..7" in c["actions"]) == (policy == "uniform")
    b5 = r["fp64_valu"]["b5_steps_per_interval"]
    assert b5["source"].startswith("device") and 1.0 <= b5["per_env_mean"] <= b5["per_wavefront_mean"] <= 16.0
    assert r["traffic"] is None and r["traffic_unit"]            # the committed PMC profile is of the physical policy (and says so when its hash matches)
    assert 1e9 < d["value"] < 1e10


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["torch.distributed.run", "self", "self-4"])
def test_bench_with_two_ranks_rehearsed_on_one_gpu(launcher):
    """The N > 1 code path of bench.py with a real world size of 2: `python -m torch.distributed.run --nproc-per-node 2 ...
    bench.py --gpus 2` (the driver's command line), both ranks on the one GPU of this box, collectives over gloo
    (SBR_BENCH_BACKEND=gloo - RCCL refuses two ranks on one device).  Everything but RCCL itself runs as in a scaling run: two
    shards by global env id, the barriers of the bracket, the all-gather inside the timed region, the MAX over ranks, ONE JSON
    line from rank 0.  Not a performance number (two ranks share a GPU) and labelled as a rehearsal."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, SBR_BENCH_BACKEND="gloo")
    for k in ("MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    ranks = 4 if launcher == "self-4" else 2           # four ranks: within the box's limit of six processes on its card
    if launcher.startswith("self"):     # plain `python bench.py --gpus N` starts its ranks itself (a child process group)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "20", "--warmup", "5"]
    else:                      # the driver's command line for N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    c = d["config"]
    assert c["self_launched"] == launcher.startswith("self")
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and c["envs_per_gpu"] == 65536 and c["envs_total"] == ranks * 65536
    assert "REHEARSAL" in c["collective_backend"] and c["allgathers_in_timed_region"] >= 1 and c["allgather_bytes_per_rank"] == 4 * 65536
    assert abs(d["value"] - ranks * 65536 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]      # whole-job aggregate over all ranks
    assert "cpu_baseline" not in d                            # rank 0 at N = 1 only
    assert ("sharded over %d GPUs" % ranks) in c["workload"] and d["roofline"]["launches_timed"] == 20
    assert d["roofline"]["larger_batches"] is None            # the in-run 262144-env leg belongs to the N = 1 line only
    # the line explains itself: how many ranks the process group had, over which backend, every rank's own
    # time and device, and the skew between them; `value` is computed from the slowest rank
    assert c["ranks"] == ranks and c["backend_reported"] == "gloo" and len(c["rank_elapsed_ms"]) == ranks and len(c["rank_devices"]) == ranks
    assert all(dv.startswith("cuda:0 ") for dv in c["rank_devices"])                 # two ranks share the one GPU of this box
    assert abs(max(c["rank_elapsed_ms"]) - d["ms_per_step"] * 20) < 1e-6 and c["rank_skew_ms"] >= 0.0


@pytest.mark.gpu
def test_numpy_rng_reset_is_the_reference_draw(G, monkeypatch):
    """code written against the reference controls an episode with `np.random.seed(k); env.reset()` - the
    influent noise is np.random.randn(48), drawn inside reset() (buffer_tank3.py:68 ... :989, gym_SBR_oneshot.py:180,
    gym_SBR_env2.py:104).  The reference-shaped classes do the same by default: seed 0 gives the reference's anchors
    (return -0.8789670883455737, Qw 0.05015591126665638), also through the gym_SBR alias of
    gym_sbr2_amd.compat, and a seeded instance (Philox on the device) leaves the global generator alone."""
    import sys
    from gym_sbr2_amd import _capi
    e = golden("sbros_const_2_5")
    assert int(e["seed"]) == 0
    np.random.seed(0)
    env = G.make("SBROS-v1")
    obs = env.reset()                                           # no rnd=, no seed=
    assert np.abs(np.r_[obs[0], obs[1]] - np.r_[e["reset_obs_DO"], e["reset_obs_EC"]]).max() < 1e-6     # RK4 fill phase: measured 2e-8
    assert np.abs(env._influent[1:] - e["influent_mixed"][1:]).max() < 1e-11          # the reference's influent of seed 0
    after = np.random.get_state()[1].copy()
    np.random.seed(0); np.random.randn(48); np.random.randn(48)                        # the reference draws twice for scenario 6
    assert np.array_equal(after, np.random.get_state()[1])                             # generator left where the reference leaves it
    total, done, k = 0.0, False, 0
    while not done:
        _, _, r, done, _ = env.step([2.0, 5.0])
        total += r; k += 1
    assert k == 463 and abs(total / -0.8789670883455737 - 1) < 1e-5
    qw = float(env._vec.ctrl_row(_capi.C_QW)[0].item())
    assert abs(qw / 0.05015591126665638 - 1) < 1e-5
    # a second episode continues the global stream, like a second reset() of the reference
    np.random.seed(0); np.random.randn(48); np.random.randn(48)
    st = np.random.get_state()
    z = [np.random.randn(48), np.random.randn(48)][1]
    np.random.set_state(st)
    env.reset()
    from gym_sbr2_amd.vec_env import load_influent_tables
    means, stds = load_influent_tables()
    want = O.OracleBatch(1).mix(means, stds, np.array([6], dtype=np.int32), z[None])
    assert np.abs(env._influent[1:] - want[0, 1:]).max() < 1e-11
    env.close()
    # env.seed(k) of the old gym API seeds that generator
    env = G.make("SBROS-v1")
    assert env.seed(0) == [0]
    obs2 = env.reset()
    assert obs2 == obs
    env.close()
    # an explicitly seeded instance draws on the device and does not touch NumPy's generator
    np.random.seed(5); before = np.random.get_state()[1].copy()
    env = G.make("SBROS-v1", seed=3)
    o3 = env.reset()
    assert np.array_equal(before, np.random.get_state()[1]) and o3 != obs
    env.close()
    # through the alias a user of the reference imports
    for name in ("gym_SBR", "gym_SBR.envs"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    from gym_sbr2_amd import compat
    compat.install_as_gym_SBR()
    from gym_SBR.envs import SbrEnv2, SbrOS
    np.random.seed(0)
    env = SbrOS()
    assert env.reset() == obs
    env.close()
    # SBR-v2: scenario 0 draws once (buffer_tank3.py:68); the golden cycles were recorded after np.random.seed(11)
    g = golden("sbrv2_cycles")
    np.random.seed(11)
    e2 = SbrEnv2()
    for c in range(2):
        s0 = e2.reset()
        s1, r1, d1, info = e2.step(g["actions"][c])
        assert np.abs(s0 - g["reset_state"][c]).max() < 1e-11 and abs(r1 - g["reward"][c]) < 1e-6
        assert np.allclose(s1, g["state"][c], rtol=1e-6) and abs(e2.diagnostics()["Qw"] / g["Qw"][c] - 1) < 1e-5
    e2.close()
    for name in ("gym_SBR", "gym_SBR.envs"):
        monkeypatch.delitem(sys.modules, name, raising=False)


@pytest.mark.gpu
def test_output_rows_equal_the_state_vector_for_both_output_types(G):
    """Every entry of the `state` rows of sbr_step against [t, x] / x_1_state (gym_SBR_oneshot.py:153) recomputed on the host
    from sbr_get_state, for float32 and float64 outputs and for both workgroup sizes.  Round 4 found the 16-byte output stores
    (inline assembly, which the compiler's hazard recognizer cannot see) corrupted once they were issued back to back: the low
    half of a store's data registers was overwritten by the next store's address arithmetic inside the two-cycle window in
    which a VMEM store of more than 64 bits still reads them.  The observation rows are covered against the oracle by the
    episode tests; this one checks ALL rows of a full-size batch."""
    from gym_sbr2_amd import _capi
    x1 = np.array([0.5, 1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10])
    for n in (64, 4096 + 37, 65536):
        for dt, tol in ((torch.float32, 2e-7), (torch.float64, 1e-14)):
            env = G.SbrOSVec(n, out_dtype=dt)
            env.reset(seed=1, scenario=(torch.arange(n, device="cuda") % 8).to(torch.int32))
            a = torch.rand(n, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
            for c in range(2):
                o, s, r, d = env.step(a)
                x, ctrl = env.get_state()
                want = np.concatenate([_np(ctrl[_capi.C_T])[:, None], _np(x).T], axis=1) / x1
                got = _np(s).astype(np.float64)
                assert np.abs(got - want).max() <= tol * (1 + np.abs(want).max()), (n, dt, c)
                assert (np.abs(got - want) <= tol * (1 + np.abs(want))).all(), (n, dt, c)
                # obs_DO[1..4] = [Xbh/2000, Xba/500, So/8, Snh/10], obs_EC[1..4] = [Ss/30, Xbh/2000, Sno/10, Snh/10] of the same x (:150-156)
                xs = _np(x).T
                wo = np.stack([xs[:, 5] / 2000, xs[:, 6] / 500, xs[:, 8] / 8, xs[:, 10] / 10], axis=1)
                we = np.stack([xs[:, 2] / 30, xs[:, 5] / 2000, xs[:, 9] / 10, xs[:, 10] / 10], axis=1)
                go = _np(o).astype(np.float64)
                assert (np.abs(go[:, 1:5] - wo) <= tol * (1 + np.abs(wo))).all() and (np.abs(go[:, 10:14] - we) <= tol * (1 + np.abs(we))).all()
            env.close()


@pytest.mark.gpu
def test_trace_record_width_and_abi_version(G):
    """SBR_NTRACE grew 28 -> 31 -> 34 over the rounds; a consumer compiled against an older header would hand
    sbr_set_trace a buffer that is too small.  The width is now an argument, a mismatch is refused, and the library reports
    the ABI version of the header it was built from."""
    import ctypes as C
    from gym_sbr2_amd import _capi
    lib = _capi.load()
    assert lib.sbr_abi_version() == _capi.ABI_VERSION == 6 and _capi.NTRACE == 36
    env = G.SbrOSVec(8)
    buf = torch.zeros(4, _capi.NTRACE, 8, dtype=torch.float64, device="cuda")
    assert lib.sbr_set_trace(env._h, buf.data_ptr(), 8, 4, 34) == -1 and b"record_width" in lib.sbr_last_error(env._h)
    assert lib.sbr_set_trace(env._h, buf.data_ptr(), 8, 4, _capi.NTRACE) == 0
    assert lib.sbr_set_trace(env._h, None, 0, 0, 0) == 0                 # switching off needs no width
    env.close()
    # the per-call lists of the reference-shaped env stay as long as the records when step() is called after `done`
    e = golden("sbros_const_2_5")
    env = G.make("SBROS-v1")
    env.reset(rnd=e["rnd"])
    for k in range(463):
        _, _, _, done, _ = env.step([2.0, 5.0])
    assert done
    env.step([2.0, 5.0]); env.step([1.0, 1.0])                          # ignored by the finished env
    t = env.trajectory()
    assert all(len(v) == 463 for v in t) and len(env._actions) == 463
    env.close()


@pytest.mark.gpu
def test_configs4_fused_rollout_at_full_size(G, tables):
    """BASELINE.json configs[4] at ITS OWN size: ONE sbr_rollout(463) launch over 65536 envs under the bench's
    physical policy (on-device Philox set-points u_DO ~ U[0, 2.5], u_EC ~ U[0, 15], influent scenarios 4..7).
      * every env finishes, all returns finite, no env near a pole;
      * 64-env handles with the same global ids (first, middle and last wavefront) are bit-identical: the fused kernel's
        arithmetic does not depend on the batch around an env;
      * a 512-env oracle sample (first, middle, last wavefronts) running the same Philox policy FREE over the whole episode:
        within 1e-6 of the gate, returns within 1e-10;
      * the returns equal the sum of rewards of sbr_step replaying the sampled actions (`actions_out`) on the sample."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n, steps, seed, pseed = 65536, 463, 1000, 77
    cfg = _capi.default_config()
    cfg.act_DO_max = 2.5                                               # what the on-device policy draws from (bench.py)
    gid = np.arange(n)
    scen = (4 + gid % 4).astype(np.int32)
    env = G.SbrOSVec(n, config=cfg)
    env.reset(seed=seed, scenario=scen)
    ret, acts = env.rollout(steps, policy_seed=pseed, return_actions=True)
    x, ctrl = env.get_state()
    xn, cn, retn = _np(x), _np(ctrl), _np(ret)
    assert np.all(cn[_capi.C_DONE] == 1) and np.all(cn[_capi.C_STEPS] == steps) and np.isfinite(retn).all() and np.isfinite(xn).all()
    st = cn[_capi.C_STATUS].astype(np.int64)
    assert np.all((st & (_capi.ST_NEAR_POLE | _capi.ST_NONFINITE)) == 0)
    assert np.array_equal(cn[_capi.C_RETURN], retn)                    # the launch covered the whole episode
    assert float(_np(acts[:, :, 0]).max()) <= 2.5 and float(_np(acts[:, :, 1]).max()) <= 15.0
    # ---- 64-env handles with the same global ids: bit for bit
    for first in (0, n // 2, n - 64):
        cfg_s = _capi.default_config(); cfg_s.act_DO_max = 2.5
        small = G.SbrOSVec(64, first_env_id=first, config=cfg_s)
        small.reset(seed=seed, scenario=scen[first:first + 64])
        rs = small.rollout(steps, policy_seed=pseed)
        xs, cs = small.get_state()
        assert torch.equal(rs, ret[first:first + 64]) and torch.equal(xs, x[:, first:first + 64]) and torch.equal(cs, ctrl[:, first:first + 64])
        small.close()
    # ---- oracle sample, free-running with the same Philox policy
    waves = [0, 1, 2, n // 128, n // 128 + 1, n // 64 - 3, n // 64 - 2, n // 64 - 1]
    pick = np.concatenate([np.arange(w * 64, w * 64 + 64) for w in waves])
    assert len(pick) == 512
    worst_g, worst_r = 0.0, 0.0
    for w in waves:
        ids = np.arange(w * 64, w * 64 + 64)
        p = O.default_params(); p.act_DO_max = 2.5
        ora = O.OracleBatch(64, params=p, nthreads=8, first_env_id=int(ids[0]))
        ora.reset(ora.mix(means, stds, scen[ids], ora.normals(seed)))
        assert np.array_equal(_np(acts[:3, ids[0]:ids[0] + 64]), ora.policy_actions(3, pseed))     # same stream, same float32 actions
        oret = ora.rollout(steps, pseed)
        g = gate(xn.T[ids], ora.envs["x"]).max()
        worst_g, worst_r = max(worst_g, g), max(worst_r, np.abs(retn[ids] - oret).max())
        assert np.array_equal(st[ids], ora.envs["status"].astype(np.int64))
        assert np.abs(cn[_capi.C_QW][ids] / ora.envs["qw"] - 1).max() < 1e-9
    print("configs[4] at 65536 envs, 512-env oracle sample free-running: worst gate %.3e, worst |d return| %.3e" % (worst_g, worst_r))
    assert worst_g < 1e-6 and worst_r < 1e-10
    # ---- sbr_step replaying the sampled actions on the first and the last wavefront
    for first in (0, n - 64):
        cfg_s = _capi.default_config(); cfg_s.act_DO_max = 2.5
        rep = G.SbrOSVec(64, first_env_id=first, config=cfg_s, out_dtype=torch.float64)
        rep.reset(seed=seed, scenario=scen[first:first + 64])
        tot = torch.zeros(64, dtype=torch.float64, device="cuda")
        for c in range(steps):
            _, _, r, _ = rep.step(acts[c, first:first + 64].contiguous())
            tot += r
        assert torch.allclose(tot, ret[first:first + 64], rtol=0, atol=1e-12)
        xr, _ = rep.get_state()
        assert gate(_np(xr).T, xn.T[first:first + 64]).max() < 1e-6
        rep.close()
    env.close()


@pytest.mark.gpu
def test_configs3_shard_of_a_fused_rollout_equals_the_slice_of_the_262144_env_rollout(G):
    """configs[3] x configs[4]: rank 3 of 8 of a 262144-env batch (32768 envs, first_env_id 98304) runs the fused rollout on its
    own; the same global ids inside ONE 262144-env rollout give the same returns, plants and controller rows, bit for bit."""
    from gym_sbr2_amd import ShardedSbrOS, _capi
    n_global, world, rank, steps = 262144, 8, 3, 463
    cfg = _capi.default_config(); cfg.act_DO_max = 2.5
    sh = ShardedSbrOS(n_global, rank=rank, world=world, device=0, config=cfg)
    n, first = sh.stop - sh.start, sh.start
    assert (n, first) == (32768, 98304)
    scen_of = lambda gid: 4 + gid % 4                                  # noqa: E731
    sh.reset(seed=1000, scenario_of=scen_of)
    r_sh = sh.rollout(steps, policy_seed=77)
    cfg_b = _capi.default_config(); cfg_b.act_DO_max = 2.5
    big = G.SbrOSVec(n_global, config=cfg_b)
    big.reset(seed=1000, scenario=scen_of(torch.arange(n_global, device="cuda")).to(torch.int32))
    r_big = big.rollout(steps, policy_seed=77)
    x_sh, c_sh = sh.env.get_state(); x_big, c_big = big.get_state()
    assert torch.equal(r_sh, r_big[first:first + n]) and torch.equal(x_sh, x_big[:, first:first + n]) and torch.equal(c_sh, c_big[:, first:first + n])
    assert bool((c_big[_capi.C_DONE] == 1).all()) and bool(torch.isfinite(r_big).all())
    sh.close(); big.close()


@pytest.mark.gpu
def test_implicit_so_sno_memories_across_every_writer(G, tables):
    """Round 4: after an ordinary step So[-1] / Sno[-1] ARE x[8] / x[9] (every interval ends with So.append(x_out[-1][8]),
    Sno.append(x_out[-1][9]), gym_SBR_oneshot.py:1955-1956, :2043-2044), and k_step neither stores nor loads their rows.  The
    public rows must still read as the reference's lists would, whoever wrote last: the reset (Ss in the Sno memory, :1652),
    an import of values that are NOT the plant's, a fused rollout in between, the done call."""
    from gym_sbr2_amd import _capi
    means, stds = tables
    n = 192
    scen = (np.arange(n) % 8).astype(np.int32)
    z = np.random.RandomState(5).randn(n, 48)
    rs = np.random.RandomState(6)
    env = G.SbrOSVec(n, out_dtype=torch.float64, action_dtype=torch.float64)
    env.reset(scenario=scen, rnd=z)
    x, ctrl = env.get_state()
    assert torch.equal(ctrl[_capi.C_SNO_M1], x[2]) and torch.equal(ctrl[_capi.C_SO_M1], x[8])          # the reset's quirk
    ora = O.OracleBatch(n, nthreads=8)
    ora.reset(ora.mix(means, stds, scen, z))

    def lockstep(calls):
        for _ in range(calls):
            a = np.column_stack([rs.uniform(0, 2.5, n), rs.uniform(0, 15, n)])
            xx, cc = env.get_state()
            ora.load_state(_np(xx), _np(cc))
            _, _, r, d = env.step(torch.from_numpy(a).cuda())
            _, _, orr, od = ora.step(a)
            xx, cc = env.get_state()
            cc = _np(cc)
            assert np.abs(_np(r) - orr).max() < 1e-11 and np.array_equal(_np(d), od)
            for row, key in ((_capi.C_SO_M1, "so_m1"), (_capi.C_SO_M2, "so_m2"), (_capi.C_SNO_M1, "sno_m1"), (_capi.C_SNO_M2, "sno_m2"),
                             (_capi.C_KLA_LAST, "kla_last"), (_capi.C_EC_LAST, "ec_last")):
                assert np.allclose(cc[row], ora.envs[key], rtol=1e-9, atol=1e-12), key
        return xx, torch.from_numpy(cc).cuda()

    # a call in which NO interval runs (t injected as NaN) straight after the reset must leave the memories where
    # they are - Sno[-1] = Ss of the fill, held in the ROWS, not x[9] - and must not claim the implicit form
    x0s, c0s = env.get_state()
    c_nan = c0s.clone(); c_nan[_capi.C_T] = float("nan")
    env.set_state(x0s, c_nan)
    env.step(torch.zeros(n, 2, dtype=torch.float64, device="cuda"))
    xn, cn = env.get_state()
    assert torch.equal(xn, x0s)                                                                          # the plant did not move
    assert torch.equal(cn[_capi.C_SNO_M1], x0s[2]) and torch.equal(cn[_capi.C_SO_M1], c0s[_capi.C_SO_M1])   # still the reset's values
    assert not torch.equal(cn[_capi.C_SNO_M1], xn[9])
    env.set_state(x0s, c0s)                        # back to the state after the reset (steps, return and time included)
    x, ctrl = lockstep(6)
    assert torch.equal(ctrl[_capi.C_SO_M1], x[8]) and torch.equal(ctrl[_capi.C_SNO_M1], x[9])          # implicit now
    # an import of memories that are not the plant's values: the next step must use THEM (the oracle gets the same rows)
    ctrl2 = ctrl.clone()
    ctrl2[_capi.C_SO_M1] += 0.37; ctrl2[_capi.C_SNO_M1] += 1.25
    env.set_state(x, ctrl2)
    x3, ctrl3 = env.get_state()
    assert torch.equal(ctrl3[_capi.C_SO_M1], ctrl2[_capi.C_SO_M1]) and torch.equal(ctrl3[_capi.C_SNO_M1], ctrl2[_capi.C_SNO_M1])
    x, ctrl = lockstep(1)                          # (the oracle was loaded with the same injected rows)
    lockstep(3)
    # a plant WITHOUT controller rows must not drag the memories along (they were x[8], x[9] of the old plant)
    x, ctrl = env.get_state()
    x_new = x.clone(); x_new[8] += 0.11; x_new[9] += 0.21
    env.set_state(x_new, None)
    x4, ctrl4 = env.get_state()
    assert torch.equal(x4, x_new) and torch.equal(ctrl4, ctrl)
    env.set_state(x, None)
    lockstep(2)
    # a fused rollout in between reads the implicit memories and leaves explicit rows behind
    x, ctrl = env.get_state()
    ref = G.SbrOSVec(n, out_dtype=torch.float64)
    ref.set_state(x, ctrl)
    _, acts = env.rollout(5, policy_seed=3, return_actions=True)
    for c in range(5):
        ref.step(acts[c])
    xa, ca = env.get_state(); xb, cb = ref.get_state()
    keep = [r_ for r_ in range(_capi.NCTRL) if r_ != _capi.C_PLAN]          # the plan row is sbr_step's report only: a rollout leaves 0
    assert gate(_np(xa).T, _np(xb).T).max() < 1e-6 and torch.allclose(ca[keep], cb[keep], rtol=1e-11, atol=1e-13)
    assert torch.equal(ca[_capi.C_SO_M1], xa[8]) and torch.equal(ca[_capi.C_SNO_M1], xa[9])
    ref.close()
    # ... and stepping on from there, through the done call (whose terminal phases move x away from the memories)
    xx, cc = env.get_state()
    calls_left = 463 - int(cc[_capi.C_STEPS, 0].item())
    x, ctrl = lockstep(calls_left)
    assert bool((ctrl[_capi.C_DONE] == 1).all())
    assert not torch.equal(ctrl[_capi.C_SO_M1], x[8])              # So after the idle phase is not the reaction phases' last So
    env.close()
