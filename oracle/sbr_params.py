"""Constants of the SBROS-v1 path (TEST INFRASTRUCTURE - part of the CPU oracle).

Every value cites where the reference defines it (paths relative to
/root/reference/gym_SBR/envs/).  tests/test_oracle_golden.py asserts the derived
ones against tests/golden/constants.npz, which was captured from the running reference.
"""
import math

import numpy as np

# ---- plant / time grid ------------------------------------------------- gym_SBR_oneshot.py:25-37
WV = 1.32
T_RATIO = (4.2 / 100, 8.3 / 100, 37.5 / 100, 31.2 / 100, 2.1 / 100, 8.3 / 100, 2.1 / 100, 6.3 / 100)
DT = 0.002 / 24
T_DELTA = DT * 10
T_CYCLE = 12 / 24

# ---- ASM1 ---------------------------------------------------------------- gym_SBR_oneshot.py:116-119
YA, YH, FP, IXB, IXP = 0.24, 0.67, 0.08, 0.08, 0.06
MUH, KS, KOH, KNO, BH, ETAG, ETAH, KH, KX, MUA, KNH, BA, KOA, KA = (
    4.0, 10.0, 0.2, 0.5, 0.3, 0.8, 0.8, 3.0, 0.1, 0.5, 1.0, 0.05, 0.4, 0.05)


def do_saturation(temp_c):
    """O2 saturation concentration, module_temperature.py:3-20."""
    tk = (temp_c + 273.15) / 100
    f = 56.12 * np.exp(-66.7354 + 87.4755 / tk + 24.4526 * np.log(tk))
    return 0.9997743214 * (8 / 10.5) * 6791.5 * f


SO_SAT = float(do_saturation(15))           # gym_SBR_oneshot.py:80 (DO_control_par[10])
KLA_MIN, KLA_MAX = 0.0, 240.0               # gym_SBR_oneshot.py:80 (DO_control_par[4:6])
KC_DO, TAUI_DO, TAUD_DO = 100.0, 20.0, 0.0  # gym_SBR_oneshot.py:83-85
EC_MIN, EC_MAX = 0.0, 0.0005                # gym_SBR_oneshot.py:89 (EC_control_par[4:6])
KC_EC, TAUI_EC, TAUD_EC = 100.0, 20.0, 0.0  # gym_SBR_oneshot.py:92-94
EC_CONC = 1200000 * 4.0                     # gym_SBR_oneshot.py:96

# ---- episode start ------------------------------------------------------- gym_SBR_oneshot.py:197-214
X0_INIT = (0.6161484733495801, 30, 0.571098000538576, 1440.01157895393, 31.254221999137,
           2599.2714348941, 168.915006750837, 551.901552960823, 2.16607843793004, 13.3791460027604,
           0.00562880208518134, 0.35996687629947, 1.86916737961228, 3.790463057094611)
IV_INIT = 0.6161484733495801
SCENARIO_DEFAULT = 6                        # gym_SBR_oneshot.py:180
U_DO_INIT, U_EC_INIT = 0.0, 15.0
ACT_DO_MAX, ACT_EC_MAX = 8.0, 15.0          # gym_SBR_oneshot.py:865-870, :901-906

# ---- observation normalisers --------------------------------------------- gym_SBR_oneshot.py:150-156
X1_STATE = (0.5, 1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10)
OBS_IDX_DO = (0, 5, 6, 8, 10)
OBS_IDX_EC = (0, 2, 5, 9, 10)
X1_DO = (0.5, 2000, 500, 8.0, 10)
X1_EC = (0.5, 30, 2000, 10, 10)
# xdot scales gym_SBR_oneshot.py:1069-1076, order as appended at :1111-1112
XDOT_DO = ((5, 4000.0), (6, 500.0), (8, 8.0), (10, 50.0))
XDOT_EC = ((2, 50.0), (5, 4000.0), (9, 50.0), (10, 50.0))

# ---- terminal phases ------------------------------------------------------ gym_SBR_oneshot.py:123-124, 2189-2218
BIOMASS_SETPOINT = 2700.0
QEFF = 0.66
SETTLER_AREA = (1.25 / 2) ** 2
SETTLER_VMAX = 474.0

# ---- mixed tolerance of the parity gate ------------------------------
STATE_SCALE = np.array(X1_STATE[1:], dtype=np.float64)
RTOL_GATE = 1e-5


def phase_times():
    """Restates module_batch_time.py:3-116 for the five scalars the path consumes.

    Each phase p starts one t_delta after the previous one ends (phase 1 starts at 0); the saved
    grid is linspace(start, end, int((end-start)/(10*t_delta))) refined by linspace(a, b,
    int((b-a)/t_delta))[1:] between consecutive nodes.  Returns (first, last, count) per phase.
    """
    out = []
    end = 0.0
    for p, ratio in enumerate(T_RATIO):
        start = end if p == 0 else end + T_DELTA
        end = start + T_CYCLE * ratio
        nodes = np.linspace(start, end, int((end - start) / (T_DELTA * 10)))
        mem = [nodes[0]]
        for a, b in zip(nodes[:-1], nodes[1:]):
            mem.extend(np.linspace(a, b, int((b - a) / T_DELTA))[1:].tolist())
        out.append((float(mem[0]), float(mem[-1]), len(mem)))
    return out


_PT = phase_times()
T1_END = _PT[0][1]     # end of fill          (0.021)
T3_0 = _PT[2][0]       # first aerobic phase starts
T3_END = _PT[2][1]
T4_END = _PT[3][1]
T5_END = _PT[4][1]     # episode's reaction part is over once t >= T5_END
