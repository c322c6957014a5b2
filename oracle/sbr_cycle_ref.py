"""CPU oracle, layer 1, for the per-cycle environment `SBR-v2` (TEST INFRASTRUCTURE).

Restates - it does not import - `SbrEnv2` (/root/reference/gym_SBR/envs/gym_SBR_env2.py:58-193), its cycle driver
`SBR_model_FB.run` (SBR_model_FB.py:8-295), the phase simulators of sub_phases_FB.py (`filling.sim_rxn` :178-271,
`rxn.sim_rxn` :406-500, `settling.sim_settling` :716-775, `drawing.sim_drawing` + `cal_eq` :780-915) and the reward
module_reward.py:4-51, with the reference's own LSODA (`integrator="lsoda"`) or the fixed-step RK4 the HIP kernel uses.
One step() = one whole 12 h cycle: five PID-controlled phases, settle, draw, aerated idle.

Pinned by tests/test_cycle_oracle.py against tests/golden/sbrv2_cycles.npz (captured from the running reference).
Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import math

import numpy as np
from scipy.integrate import odeint

from . import sbr_params as P
from .sbr_ref import conversion, influent_mix, rk4, settle_closed_form

T_DELTA = 0.002 / 24                         # gym_SBR_env2.py:34, SBR_model_FB.py:29
# DO_control_par of gym_SBR_env2.py:48: Kc, tauI, delt, (set-point), Kla_min, Kla_max, ..., tauD = [9], So_sat = [10]
KC, TAUI, DT_PID, KLA_MIN, KLA_MAX, TAUD = 5.0, 0.00035, 0.02 / 24, 0.0, 240.0, 0.005
QEFF, BIOMASS_SETPOINT = 0.66, 2700          # SBR_model_FB.py:216-218
SCENARIO = 0                                 # gym_SBR_env2.py:104


def rhs_fill(x, t, kla, loading):
    """filling.dxdt, sub_phases_FB.py:42-176 (no in-place block in this file)."""
    r = conversion(x, kla)
    d = np.zeros(14)
    d[0] = loading[0]
    for i in range(1, 14):
        d[i] = r[i] + (loading[0] / x[0]) * (loading[i] - x[i])
    return d


def rhs_rxn(x, t, kla):
    """rxn.dxdt, sub_phases_FB.py:278-404: conversion only, volume constant."""
    r = conversion(x, kla)
    d = np.zeros(14)
    d[1:] = r[1:]
    return d


class SbrEnv2Ref:
    def __init__(self, tables, integrator="lsoda"):
        self.tables, self.integrator = tables, integrator

    # ------------------------------------------------------------------ one PID-controlled phase
    def _phase(self, t_start, t_end, x, sp, kla_in, loading=None):
        """filling.sim_rxn / rxn.sim_rxn: control intervals t_save2, positional PID with bias Kla[0].
        Quirk kept: interval 0 OVERWRITES Kla[0] (which held the incoming bias), so the bias of intervals 1.. is the
        controlled, clamped value of interval 0 (sub_phases_FB.py:219,243 / :447,465)."""
        n2 = int((t_end - t_start) / (T_DELTA * 10))
        grid = np.linspace(t_start, t_end, n2)
        n_iv = n2 - 1
        so = np.zeros(n_iv)
        ie = np.zeros(n_iv)
        kla = np.zeros(n_iv)
        so[0] = x[8]
        kla[0] = kla_in
        x = np.array(x, dtype=np.float64)
        for i in range(n_iv):
            t_range = np.linspace(grid[i], grid[i + 1], int((grid[i + 1] - grid[i]) / T_DELTA))
            e = sp - so[i]
            dcv = 0.0
            if i >= 1:
                dcv = (so[i] - so[i - 1]) / DT_PID
                ie[i] = ie[i - 1] + e * DT_PID
            k = KC * e + KC / TAUI * ie[i] + KC * TAUD * dcv + kla[0]
            if k > KLA_MAX:
                k = KLA_MAX
                ie[i] = ie[i] - e * DT_PID
            if k < KLA_MIN:
                k = KLA_MIN
                ie[i] = ie[i] - e * DT_PID
            kla[i] = k
            if self.integrator == "lsoda":
                if loading is not None:
                    x = odeint(rhs_fill, x, t_range, args=(k, loading))[-1]
                else:
                    x = odeint(rhs_rxn, x, t_range, args=(k,))[-1]
            else:
                if loading is not None:
                    x = rk4(rhs_fill, x, grid[i], grid[i + 1], 10, (k, loading))
                else:
                    x = rk4(rhs_rxn, x, grid[i], grid[i + 1], 10, (k,))
            if i < n_iv - 1:
                so[i + 1] = x[8]
        return x, kla

    # ------------------------------------------------------------------ the gym surface
    def reset(self, rnd, scenario=SCENARIO):
        means, stds = self.tables
        self.influent = influent_mix(means[scenario], stds[scenario], np.asarray(rnd, dtype=np.float64))
        self.x0 = np.array(P.X0_INIT, dtype=np.float64)
        self.iv = P.IV_INIT
        self.qin = P.WV - self.iv
        tot = self.x0 + self.influent                         # gym_SBR_env2.py:108-119
        cod = tot[1] + tot[2] + tot[3] + tot[4] + tot[5] + tot[6] + tot[7]
        return np.array([tot[0], (cod - 5145) / 10, tot[10] / 30])

    def step(self, action):
        a = np.clip(np.asarray(action, dtype=np.float64), 0.0, 1.0)      # :133
        sp = [0, 0, a[0] * 8, 0, a[1] * 8, 0, 0, a[2] * 8]               # :184-186 on DO_setpoints = [0,0,2,0,2,0,0,2]
        t_ph = [P.T_CYCLE * r for r in P.T_RATIO]
        loading = self.influent.copy()
        loading[0] = self.qin / (P.T_CYCLE * P.T_RATIO[0])               # :144
        self.phases = []
        # phase 1 (fill) .. 5
        t_start, t_end = 0.0, 0.0 + t_ph[0]
        x, kla = self._phase(t_start, t_end, self.x0, sp[0], 0.0, loading)
        self.phases.append((x, kla))
        for ph in (1, 2, 3, 4):
            t_start = t_end + T_DELTA
            t_end = t_start + t_ph[ph]
            x, kla = self._phase(t_start, t_end, x, sp[ph], kla[-1])
            self.phases.append((x, kla))
        x5, kla3, kla5 = x, self.phases[2][1], self.phases[4][1]
        # phase 6: settle (sub_phases_FB.py:716-775; v == vmax => linear layer system)
        t_start = t_end + T_DELTA
        t_end = t_start + t_ph[5]
        xf = 0.75 * (x5[3] + x5[4] + x5[5] + x5[6] + x5[7])
        z = x5[0] / P.SETTLER_AREA
        if self.integrator == "lsoda":
            def f(sx, tt):
                j = P.SETTLER_VMAX * sx
                d = np.empty(10)
                d[0] = j[1] / z
                d[1:9] = (j[2:10] - j[1:9]) / z
                d[9] = (0 - j[9]) / z
                return d
            sx = odeint(f, [xf] * 10, np.linspace(t_start, t_end, int((t_end - t_start) / T_DELTA)))[-1]
        else:
            sx = settle_closed_form(xf, P.SETTLER_VMAX / z * (t_end - t_start))
        self.sx, self.xf = sx.copy(), xf
        # phase 7: draw + wastage + effluent quality (sub_phases_FB.py:780-915)
        t_start = t_end + T_DELTA
        t_end = t_start + t_ph[6]
        layer_v = x5[0] / 10
        resid_v = x5[0] - QEFF
        m = int(math.ceil(round(QEFF / layer_v)))
        sx_eff = sum(sx[-m:-1] * layer_v)
        xe = np.array(x5, dtype=np.float64)
        xe[0] = QEFF
        for i in (4, 7, 3, 5, 6):
            xe[i] = xe[i] * (1 / 0.75) * sx_eff / xf
        w = layer_v * sx[0:10 - m]
        rs = sx[0:10 - m].copy()
        waste = sum(w) - BIOMASS_SETPOINT * resid_v
        qw = float("nan")
        for i in range(10 - m):
            rest = waste - w[i]
            if rest > 0:
                waste = rest
                rs[i] = 0
                w[i] = 0
                resid_v -= layer_v
            else:
                qw = waste / (rs[i] - BIOMASS_SETPOINT)
                w[i] = w[i] - qw * rs[i]
                resid_v -= qw
                rs[i] = w[i] / (layer_v - qw)
                break
        sx2 = sum(w) / resid_v
        x7 = np.array(x5, dtype=np.float64)
        x7[0] = resid_v
        for i in (4, 7, 3, 5, 6):
            x7[i] = x5[i] * (1 / 0.75) * sx2 / xf
        # cal_eq on the effluent composition
        snkj = xe[10] + xe[11] + xe[12] + 0.08 * (xe[5] + xe[6]) + 0.06 * (xe[7] + xe[3])
        ntot = xe[9] + snkj
        ss_ = 0.75 * (xe[4] + xe[3] + xe[5] + xe[6] + xe[7])
        bod5 = 0.25 * (xe[2] + xe[4] + (1 - 0.08) * (xe[5] + xe[6]))
        cod = xe[2] + xe[1] + xe[4] + xe[3] + xe[5] + xe[6] + xe[7]
        eqi = (2 * ss_ + 1 * cod + 30 * snkj + 10 * xe[9] + 2 * bod5) * (1 / 1000) * 0.66
        self.eff = np.array([0.66, ntot, cod, xe[10], bod5, xe[9]])
        self.qw, self.eqi, self.x_after_draw = qw, eqi, x7
        # phase 8: aerated idle, from the drawn reactor, bias = last Kla of phase 5 (SBR_model_FB.py:258)
        t_start = t_end + T_DELTA
        t_end = t_start + t_ph[7]
        x8, kla8 = self._phase(t_start, t_end, x7, sp[7], kla5[-1])
        self.phases.append((x8, kla8))
        self.x_last = x8
        # reward, module_reward.py:4-51
        td = 0.002 / 24
        ae3 = 1.32 * sum(kla3) * td / (len(kla3) * td)
        ae5 = 1.32 * sum(kla5) * td / (len(kla5) * td)
        ae8 = (1.32 - qw) * sum(kla8) * td / (len(kla8) * td)
        ae = P.SO_SAT / (1.8 * 1000) * (ae3 + ae5 + ae8)
        pe = (0.004 * self.qin + 0.05 * qw + 0.004 * QEFF)
        me = 0.005 * 1.32 * 24 + 0.005 * 1.32 * 24
        oci = ae + pe + me
        reward = (5 - oci) + (0 if self.eff[3] < 4 else -20)
        self.oci = oci
        state = np.array([QEFF, self.eff[2], self.eff[3] / 30])
        return state, reward, True, {}
