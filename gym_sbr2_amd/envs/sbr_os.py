"""`SbrOS` - the reference's single-environment class (gym_SBR/envs/gym_SBR_oneshot.py:99,
id `SBROS-v1`) with the same surface, backed by the batched HIP environment at N = 1.

    reset()                      -> (obs_DO: list[9], obs_EC: list[9])                     (:438)
    step([u_DO, u_EC])           -> (obs, state: ndarray[15], reward: float, done: bool, {})  (:1273)
    get_available_actions(...)   -> [ndarray, ndarray]                                     (:440-459)

The class derives from `gym.Env` when gym is importable (as upstream, :99), else from `gymnasium.Env`, else from `object`
(`gym_sbr2_amd._gymcompat`); either way it speaks the reference's OLD gym API generation: reset() returns the observation only
and step() the reference's own 5-tuple.

Outputs are float64 like the reference's.  Differences, all deliberate: state lives on the GPU
instead of module globals (so several instances can coexist), the influent noise comes from an
explicit `seed`/`rnd` instead of the global numpy RNG, and the 18 lists of `trajectory()` are
sampled once per step() call (see its docstring).
"""
import numpy as np
import torch

from .. import _gymcompat as _gym
from ..vec_env import SbrOSVec

_Box = _gym._Box          # (kept for callers that imported the stand-in from here)


def _hermite(nodes, slopes, span, tau):
    """Cubic Hermite interpolation of the S + 1 equidistant nodes [S+1, 14] with slopes [S+1, 14] over [0, span] at tau [m]."""
    s_n = nodes.shape[0] - 1
    h = span / s_n
    k = np.minimum((tau / h).astype(int), s_n - 1)
    th = ((tau - k * h) / h)[:, None]
    h00, h10 = 2 * th ** 3 - 3 * th ** 2 + 1, th ** 3 - 2 * th ** 2 + th
    h01, h11 = -2 * th ** 3 + 3 * th ** 2, th ** 3 - th ** 2
    return h00 * nodes[k] + h10 * h * slopes[k] + h01 * nodes[k + 1] + h11 * h * slopes[k + 1]


class SbrOS(_gym.Env):
    metadata = {"render.modes": ["human"]}                      # :101

    def __init__(self, device=0, seed=None, reward=None):
        # the reference declares stale spaces (:106-113); these are the real ones of step()
        self.action_space = _gym.box([0.0, 0.0], [8.0, 15.0])
        self.observation_space = _gym.box(np.full(18, -np.inf), np.full(18, np.inf))
        # reward: None / "eqi_oci" = the reference's (module_reward_EQIOCI.py); "g2anet", "oci" = the other reward modules
        self._vec = SbrOSVec(1, device=device, out_dtype=torch.float64, action_dtype=torch.float64, reward=reward)
        self._vec.enable_host_io()        # the kernel reads the action from, and writes its outputs to, pinned host memory
        self._seed = seed
        self._episodes = 0
        self._rewards, self._states, self._actions = [], [], []
        self._x_postfill = None

    def seed(self, seed=None):
        self._seed = seed
        return [seed]

    def _split(self, obs_row):
        o = obs_row.tolist()
        return o[:9], o[9:]

    def reset(self, rnd=None, scenario=None, influent=None, carry_over=False):
        """rnd: the 48 standard normals buffer_tank3.py:68 would draw (None: drawn on the device from
        `seed` + episode count); scenario: 0..7 (None = 6, as the reference :180); carry_over: start the new cycle
        from the state the last one ended in (the reference's disabled x0_new / IV_new, :260-268)."""
        seed = (0 if self._seed is None else int(self._seed)) + self._episodes
        self._episodes += 1
        self._rewards, self._states, self._actions = [], [], []
        self._trace = self._vec.enable_trace(1, 464)
        obs = self._vec.reset(seed=seed, carry_over=carry_over,
                              scenario=None if scenario is None else [int(scenario)],
                              rnd=None if rnd is None else np.asarray(rnd, dtype=np.float64)[None],
                              influent=None if influent is None else np.asarray(influent, dtype=np.float64)[None])
        self._x_postfill = self._vec.get_state()[0][:, 0].cpu().numpy()       # where the first interval starts (dense trajectory)
        return self._split(obs[0].cpu())

    def step(self, action):
        obs, state, reward, done = self._vec.step_host([[float(action[0]), float(action[1])]])
        state = state[0].copy()
        reward, done = float(reward[0]), bool(done[0])
        self._rewards.append(reward)
        self._states.append(state)
        self._actions.append((float(action[0]), float(action[1])))
        return self._split(obs[0]), state, reward, done, {}

    def get_available_actions(self, pre_action, n_agents, n_action):
        """Mask of the discrete set-point moves that stay inside the action bounds (:440-459)."""
        deltas = ([-0.1, 0, 0.1], [-5, 0, 5])
        bounds = ([0, 8], [0, 15])
        out = []
        for agent in range(n_agents):
            ok = np.ones(n_action)
            for i in range(n_action):
                v = pre_action[agent] + deltas[agent][i]
                ok[i] = 1 if bounds[agent][0] <= v <= bounds[agent][1] else 0
            out.append(ok)
        return out

    def _dense_rows(self, rec):
        """The reference's sub-interval rows of the running episode: for every control interval the solution on ITS output grid
        t_range = linspace(t, t + t_delta, int(((t + t_delta) - t)/dt)) (:1339, :1384: 9 or 10 points with the rounding of the
        span), from which it appends t_range[1:] to t_t, x_out[1:] to x_t, x_out[:-1, k] to So_t / Ss_t / Sno_t / Snh_t and
        len - 1 copies of the set-points to u_DO_t / u_EC_t (:1359-1369, :876-892).  LSODA interpolates its own steps onto that
        grid; here the RK4 nodes of every interval and the right-hand side at them come from the device (sbr_eval_substeps,
        replayed from the recorded start state, Kla and EC of the interval) and are interpolated by cubic Hermite polynomials
        - fourth order like the integrator, exact at the nodes, so the last row of an interval is the state step() returned."""
        from .. import _capi as K
        cfg = self._vec.cfg
        dt, t_delta, n = cfg.dt, cfg.t_delta, rec.shape[0]
        n_iv = rec[:, K.TR_N_IV].astype(int)
        t_end = rec[:, K.TR_T]
        t0 = np.concatenate([[cfg.T_fill], t_end[:-1]])
        x0 = np.vstack([self._x_postfill[None], rec[:-1, K.TR_X0:K.TR_X0 + 14]])
        first_kla, first_ec = rec[:, K.TR_KLA_FIRST], rec[:, K.TR_EC_FIRST]

        def span_of(t):
            return (t + t_delta) - t

        def setpoints(t, action):          # the phase tests and clipping of step() (:860-906) for an interval that starts at t
            aerobic = (cfg.T3_0 <= t <= cfg.T3_end) or t > cfg.T4_end
            a0, a1 = min(max(action[0], 0.0), cfg.act_DO_max), min(max(action[1], 0.0), cfg.act_EC_max)
            return (a0, 0.0) if aerobic else (0.0, a1)

        live = np.nonzero(n_iv >= 1)[0]
        xs1, dx1 = (v.cpu().numpy() for v in self._vec.eval_substeps(x0[live], first_kla[live], first_ec[live], span_of(t0[live])))
        two = np.nonzero(n_iv[live] >= 2)[0]
        if len(two):
            t_mid = t0[live][two] + t_delta
            xs2, dx2 = (v.cpu().numpy() for v in self._vec.eval_substeps(xs1[two, -1], rec[live][two, K.TR_KLA],
                                                                          rec[live][two, K.TR_EC], span_of(t_mid)))
        out = {k: [] for k in ("t_t", "x_t", "So_t", "Ss_t", "Sno_t", "Snh_t", "u_DO_t", "u_EC_t")}

        def emit(t_start, nodes, slopes, u_do, u_ec):
            span = span_of(t_start)
            grid = np.linspace(t_start, t_start + t_delta, int(span / dt))          # the reference's t_range, bit for bit
            rows = _hermite(nodes, slopes, span, np.clip(grid - t_start, 0.0, span))
            rows[0], rows[-1] = nodes[0], nodes[-1]
            out["t_t"] += grid[1:].tolist()
            out["x_t"].append(rows[1:])
            for name, j in (("So_t", 8), ("Ss_t", 2), ("Sno_t", 9), ("Snh_t", 10)):
                out[name] += rows[:-1, j].tolist()
            out["u_DO_t"] += [u_do] * (len(grid) - 1)
            out["u_EC_t"] += [u_ec] * (len(grid) - 1)

        second = {int(live[j]): i for i, j in enumerate(two)}
        for i, k in enumerate(live):
            k = int(k)
            if k in second:
                u = setpoints(t0[k], self._actions[k])
                emit(t0[k], xs1[i], dx1[i], *u)
                j = second[k]
                emit(t0[k] + t_delta, xs2[j], dx2[j], rec[k, K.TR_U_DO], rec[k, K.TR_U_EC])
            else:
                emit(t0[k], xs1[i], dx1[i], rec[k, K.TR_U_DO], rec[k, K.TR_U_EC])
        out["x_t"] = np.vstack(out["x_t"]) if out["x_t"] else np.empty((0, 14))
        return out

    def trajectory(self, as_dict=False, dense=False):
        """The reference's 18-tuple, in its order (gym_SBR_oneshot.py:1288):

            t_t, x_t, u_DO_t, u_EC_t, state_t, So_t, Ss_t, EC, Sno_t, dcv_EC, ie_EC, e_EC,
            reward_t, reward_EQI_t, reward_OCI_t, reward_AE_t, reward_EC_t, Snh_t

        Every list has ONE entry per step() call of the running episode, recorded on the device at the end of the call
        (for a call that runs two control intervals: of the second, like the reward).  The reference grows t_t, x_t, the
        four concentration lists and the two set-point lists by the 8 or 9 rows of LSODA's output grid per interval, and its
        controller lists (EC, dcv_EC, ie_EC, e_EC) also hold the entries of the fill phase; the fixed-step integrator has
        no such grid, so those lists are sampled per call here.  reward_t and the four reward diagnostics
        (module_reward_EQIOCI.py:109-112) are per call in the reference too.  state_t, which the reference leaves empty
        (its append is commented out, :436), holds the state vector step() returned.
        as_dict=True returns the same arrays by name (plus Kla, the DO controller's output).
        dense=True returns t_t, x_t, So_t, Ss_t, Sno_t, Snh_t, u_DO_t and u_EC_t on the reference's sub-interval grid instead
        (8 or 9 rows per control interval, `_dense_rows`): the entries the reference's lists hold for the reaction intervals,
        i.e. without the 252 rows of the fill phase in front and the rows of settle / draw / idle at the end."""
        from .. import _capi as K
        n = len(self._rewards)
        rec = self._trace[:n].cpu().numpy()[:, :, 0]
        x_t = rec[:, K.TR_X0:K.TR_X0 + 14]
        cols = {"t_t": rec[:, K.TR_T], "x_t": x_t, "u_DO_t": rec[:, K.TR_U_DO], "u_EC_t": rec[:, K.TR_U_EC],
                "state_t": [s.copy() for s in self._states], "So_t": x_t[:, 8], "Ss_t": x_t[:, 2], "EC": rec[:, K.TR_EC],
                "Sno_t": x_t[:, 9], "dcv_EC": rec[:, K.TR_DCV_EC], "ie_EC": rec[:, K.TR_IE_EC], "e_EC": rec[:, K.TR_E_EC],
                "reward_t": rec[:, K.TR_REWARD], "reward_EQI_t": rec[:, K.TR_R_EQI], "reward_OCI_t": rec[:, K.TR_R_OCI],
                "reward_AE_t": rec[:, K.TR_R_AE], "reward_EC_t": rec[:, K.TR_R_EC], "Snh_t": x_t[:, 10]}
        if dense:
            cols.update(self._dense_rows(rec))
        if as_dict:
            cols["Kla"] = rec[:, K.TR_KLA]
            return cols
        order = ("t_t x_t u_DO_t u_EC_t state_t So_t Ss_t EC Sno_t dcv_EC ie_EC e_EC reward_t reward_EQI_t reward_OCI_t "
                 "reward_AE_t reward_EC_t Snh_t").split()
        return tuple(cols[k] if k in ("x_t", "state_t") or isinstance(cols[k], list) else cols[k].tolist() for k in order)

    def render(self, mode="human"):
        return None

    def close(self):
        self._vec.close()
