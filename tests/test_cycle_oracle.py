"""Pins the oracle for the per-cycle env `SBR-v2` against tests/golden/sbrv2_cycles.npz (five cycles of the reference's
SbrEnv2, captured by oracle/gen_golden.py: per phase the incoming state and bias, every interval's Kla, the end state;
settler, draw, effluent quality, observation and reward)."""
import numpy as np
import pytest
from conftest import gate, golden

from oracle import sbr_oracle as O
from oracle.sbr_cycle_ref import SbrEnv2Ref
from oracle.sbr_ref import influent_mix


FIXTURES = ["sbrv2_cycles", "sbrv2_cycles_heldout"]      # the second (round 6): eight held-out cycles, set-points inside the oxygen knee


@pytest.mark.parametrize("fixture", FIXTURES)
def test_layer1_lsoda_is_bit_identical_to_reference(tables, fixture):
    g = golden(fixture)
    env = SbrEnv2Ref(tables)
    assert g["ph_n_intervals"][:6].tolist() == [24, 48, 223, 186, 11, 36]
    for c in range(len(g["actions"])):
        st0 = env.reset(g["rnd"][c])
        st, r, done, info = env.step(g["actions"][c])
        f = int(g["phase_first"][c])
        assert np.array_equal(st0, g["reset_state"][c]) and done is True and info == {}
        for k in range(6):                                   # phases 1-5 and 8: end state and every interval's Kla
            x_end, kla = env.phases[k]
            assert np.array_equal(x_end, g["ph_x_end"][f + k]) and len(kla) == g["ph_n_intervals"][f + k]
            assert np.array_equal(kla, g["ph_Kla"][f + k][:len(kla)])
        assert np.array_equal(env.sx, g["sX"][c]) and env.qw == g["Qw"][c] and env.eqi == g["EQI"][c]
        assert np.array_equal(env.eff, g["eff"][c]) and np.array_equal(env.x_after_draw, g["x_after_draw"][c])
        assert np.array_equal(st, g["state"][c]) and r == g["reward"][c]
    if fixture == "sbrv2_cycles_heldout":
        assert len(g["actions"]) == 8 and g["reward"][0] < -10 and (g["actions"][:5] * 8 < 1.0).any(axis=1).all()    # set-points in the knee
        return
    assert g["reward"][1] < -10 and g["reward"][0] > 0       # the ammonia penalty (no aeration) is exercised
    # clipping (gym_SBR_env2.py:133): the fifth cycle was run with [1.7, -0.3, 0.5]; [1, 0, 0.5] must give the same cycle
    assert g["actions"][4].tolist() == [1.7, -0.3, 0.5]
    env.reset(g["rnd"][4])
    st, r, _, _ = env.step([1.0, 0.0, 0.5])
    assert np.array_equal(st, g["state"][4]) and r == g["reward"][4]


@pytest.mark.parametrize("fixture", FIXTURES)
def test_layer2_c_rk4_is_bit_identical_to_layer1_rk4(tables, fixture):
    means, stds = tables
    g = golden(fixture)
    py = SbrEnv2Ref(tables, integrator="rk4")
    for c in range(len(g["actions"])):
        st0 = py.reset(g["rnd"][c])
        pst, pr, _, _ = py.step(g["actions"][c])
        b = O.OracleCycleBatch(1, O.default_params(scheme=0))
        cst0 = b.reset(influent_mix(means[0], stds[0], g["rnd"][c])[None])
        st, r, diag, log = b.step_logged(0, g["actions"][c])
        assert np.array_equal(cst0[0], st0) and np.array_equal(st, pst) and r == pr
        for k in range(6):
            kla = py.phases[k][1]
            assert np.array_equal(kla, log[k][:len(kla)]) and np.isnan(log[k][len(kla):]).all()
        assert np.array_equal(b.x[0], py.x_last) and diag[0] == py.qw and diag[1] == py.eqi and diag[2] == py.oci
        assert np.array_equal(diag[3:8], py.eff[1:])


@pytest.mark.parametrize("fixture", FIXTURES)
@pytest.mark.parametrize("scheme", [0, 1])
def test_rk4_cycle_inside_gate_of_reference(tables, scheme, fixture):
    """RK4 with 10 substeps per control interval (scheme 0) and the adaptive Butcher-5 of round 5 (scheme 1: every interval) against the reference's LSODA, closed loop over the whole cycle.
    Measured: phase-end states <= 0.018 of the gate, rewards within 3.2e-8, Qw within 1.7e-7 relative (scheme 0)."""
    means, stds = tables
    g = golden(fixture)
    n = len(g["actions"])
    b = O.OracleCycleBatch(n, O.default_params(scheme=scheme), nthreads=2)
    b.reset(np.stack([influent_mix(means[0], stds[0], g["rnd"][c]) for c in range(n)]))
    st, r, diag = b.step(g["actions"])
    last = g["ph_x_end"][g["phase_first"] + 5]
    print("[info] %s, cfg.scheme = %d: cycle-end state worst %.4f of the gate, rewards within %.1e" % (fixture, scheme, gate(b.x, last).max(), np.abs(r - g["reward"]).max()))
    assert gate(b.x, last).max() <= 0.1
    assert np.abs(r - g["reward"]).max() < 1e-6 and np.abs(diag[:, 0] / g["Qw"] - 1).max() < 1e-5
    assert np.allclose(st, g["state"], rtol=1e-6, atol=1e-7) and np.abs(diag[:, 1] / g["EQI"] - 1).max() < 1e-6
    # carry-over: a second cycle from the end state of the first differs from a fresh one
    first = b.x.copy()
    b.reset(b.influent, carry_over=True)
    st2, r2, _ = b.step(g["actions"])
    assert not np.allclose(b.x, first) and np.isfinite(b.x).all() and not np.allclose(r2, r)
