"""`SbrOS` - the reference's single-environment class (gym_SBR/envs/gym_SBR_oneshot.py:99,
id `SBROS-v1`) with the same surface, backed by the batched HIP environment at N = 1.

    reset()                      -> (obs_DO: list[9], obs_EC: list[9])                     (:438)
    step([u_DO, u_EC])           -> (obs, state: ndarray[15], reward: float, done: bool, {})  (:1273)
    get_available_actions(...)   -> [ndarray, ndarray]                                     (:440-459)

The class derives from `gym.Env` when gym is importable (as upstream, :99), else from `gymnasium.Env`, else from `object`
(`gym_sbr2_amd._gymcompat`); either way it speaks the reference's OLD gym API generation: reset() returns the observation only
and step() the reference's own 5-tuple.

Outputs are float64 like the reference's.  State lives on the GPU instead of module globals (so several instances can
coexist).  The influent noise of `reset()` comes from where the reference takes it - `np.random.randn(48)` on the host
(buffer_tank3.py:68, drawn inside reset(), gym_SBR_oneshot.py:180) - unless the instance was given a `seed` (then a Philox
stream on the device, keyed by seed + episode count) or the caller passes `rnd` / `influent`: code written against the
reference controls an episode with `np.random.seed(k); env.reset()`, and that gives the reference's plant of seed k here
too.  The 18 lists of `trajectory()` hold one entry per step() call, or with `dense=True` the reference's own entries.
"""
import numpy as np
import torch

from .. import _capi
from .. import _gymcompat as _gym
from ..vec_env import SbrOSVec, _raw_stream

_Box = _gym._Box          # (kept for callers that imported the stand-in from here)


def _hermite(nodes, slopes, span, tau):
    """Hermite interpolation of the S + 1 equidistant nodes [S+1, 14] with slopes [S+1, 14] over [0, span] at tau [m]: the
    quintic through the values and slopes of the THREE nodes nearest to each tau (error O(h^6), well below the integrator's
    own; the cubic between two nodes, O(h^4), would add up to half a parity gate to the rows between the nodes).  With u the
    distance to the middle node in substeps and l_i the Lagrange polynomials of the nodes -1, 0, 1:
    y = sum_i y_i (1 - 2 l_i'(u_i)(u - u_i)) l_i(u)^2 + h y'_i (u - u_i) l_i(u)^2."""
    s_n = nodes.shape[0] - 1
    h = span / s_n
    if s_n < 2:                                  # two nodes only: the cubic
        th = (tau / h)[:, None]
        h00, h10, h01, h11 = 2 * th ** 3 - 3 * th ** 2 + 1, th ** 3 - 2 * th ** 2 + th, -2 * th ** 3 + 3 * th ** 2, th ** 3 - th ** 2
        return h00 * nodes[0] + h10 * h * slopes[0] + h01 * nodes[1] + h11 * h * slopes[1]
    m = np.clip(np.rint(tau / h).astype(int), 1, s_n - 1)
    u = (tau / h - m)[:, None]
    lm, l0, lp = u * (u - 1) / 2, 1 - u * u, u * (u + 1) / 2
    out = (1 + 3 * (u + 1)) * lm ** 2 * nodes[m - 1] + l0 ** 2 * nodes[m] + (1 - 3 * (u - 1)) * lp ** 2 * nodes[m + 1]
    return out + h * ((u + 1) * lm ** 2 * slopes[m - 1] + u * l0 ** 2 * slopes[m] + (u - 1) * lp ** 2 * slopes[m + 1])


def reference_randn(scenario):
    """The 48 standard normals of the influent draw, taken from the global NumPy generator exactly as the reference's
    buffer_tank() consumes it: every scenario block except 0 calls np.random.randn(48) TWICE and uses the second vector
    (buffer_tank3.py:206 + :224, ..., :971 + :989 for scenario 6, which SbrOS.reset uses; scenario 0, which SbrEnv2.reset uses,
    draws once, :68).  After np.random.seed(k) the generator is therefore left where the reference leaves it."""
    rnd = np.random.randn(48)
    if int(scenario) != 0:
        rnd = np.random.randn(48)
    return rnd


class SbrOS(_gym.Env):
    metadata = {"render.modes": ["human"]}                      # :101

    def __init__(self, device=0, seed=None, reward=None, rng=None, scheme=None):
        """rng: where reset() takes the 48 standard normals of the influent draw from when the caller passes neither `rnd` nor
        `influent`.  "numpy" - `np.random.randn(48)` on the host, once per reset, exactly the reference's draw
        (buffer_tank3.py:68): `np.random.seed(k)` before reset() selects the episode, as with the reference.  "philox" - drawn on
        the device from `seed` + episode count.  Default: "numpy", or "philox" when a `seed` is given.
        scheme: cfg.scheme of the library (None = its default, 1: adaptive Butcher-5 per interval; 0: ten RK4 substeps - the scheme
        whose nodes trajectory(dense=True) replays, so that its dense rows ARE the integrator's own intermediate states)."""
        # the reference declares stale spaces (:106-113); these are the real ones of step()
        self.action_space = _gym.box([0.0, 0.0], [8.0, 15.0])
        self.observation_space = _gym.box(np.full(18, -np.inf), np.full(18, np.inf))
        # reward: None / "eqi_oci" = the reference's (module_reward_EQIOCI.py); "g2anet", "oci" = the other reward modules
        cfg = None
        if scheme is not None:
            from .. import _capi
            cfg = _capi.default_config()
            cfg.scheme = int(scheme)
        self._vec = SbrOSVec(1, device=device, out_dtype=torch.float64, action_dtype=torch.float64, reward=reward, config=cfg)
        views = self._vec.enable_host_io()        # the kernel reads the action from, and writes its outputs to, pinned host memory
        self._act_row, self._obs_row, self._state_row = views[0][0], views[1][0], views[2][0]
        self._reward_view, self._done_view = views[3], views[4]
        if rng is None:
            rng = "philox" if seed is not None else "numpy"
        if rng not in ("numpy", "philox"):
            raise ValueError('rng must be "numpy" or "philox"')
        self._rng = rng
        self._seed = seed
        self._episodes = 0
        self._done = False
        self._rewards, self._states, self._actions = [], [], []
        self._start = None                        # device copies of what the dense trajectory export needs from reset()

    def seed(self, seed=None):
        """Old-gym `env.seed(k)`: seeds the generator reset() draws from - the global NumPy one for rng="numpy" (what a user of
        the reference does by hand), the instance's Philox key otherwise."""
        if self._rng == "numpy":
            np.random.seed(seed)
        else:
            self._seed = seed
        return [seed]

    def _split(self, obs_row):
        o = obs_row.tolist()
        return o[:9], o[9:]

    def reset(self, rnd=None, scenario=None, influent=None, carry_over=False):
        """rnd: the 48 standard normals buffer_tank3.py:68 would draw (None: drawn as the instance's `rng` says - by default
        np.random.randn(48), like the reference); scenario: 0..7 (None = 6, as the reference :180); carry_over: start the new
        cycle from the state the last one ended in (the reference's disabled x0_new / IV_new, :260-268)."""
        seed = (0 if self._seed is None else int(self._seed)) + self._episodes
        self._episodes += 1
        if rnd is None and influent is None and self._rng == "numpy":
            rnd = reference_randn(6 if scenario is None else int(scenario))
        self._rewards, self._states, self._actions = [], [], []
        self._done = False
        self._trace = self._vec.enable_trace(1, 464)
        x_pre = self._vec.get_state()[0] if carry_over else None      # where the fill phase starts (device tensor, no host sync)
        obs = self._vec.reset(seed=seed, carry_over=carry_over,
                              scenario=None if scenario is None else [int(scenario)],
                              rnd=None if rnd is None else np.asarray(rnd, dtype=np.float64)[None],
                              influent=None if influent is None else np.asarray(influent, dtype=np.float64)[None])
        # what trajectory(dense=True) replays from: kept as device tensors (two asynchronous copies on the launch stream) and
        # only brought to the host if a dense trajectory is ever asked for
        self._start = (x_pre, self._vec.get_state()[0], self._vec.influent())
        return self._split(obs[0].cpu())

    @property
    def _x_prefill(self):
        x_pre = self._start[0]
        return np.array(list(self._vec.cfg.x0), dtype=np.float64) if x_pre is None else x_pre[:, 0].cpu().numpy()

    @property
    def _x_postfill(self):
        return self._start[1][:, 0].cpu().numpy()       # where the first interval starts

    @property
    def _influent(self):
        return self._start[2][:, 0].cpu().numpy()       # the loading vector, [0] = inflow during the fill

    def step(self, action):
        v = self._vec
        self._act_row[0] = action[0]; self._act_row[1] = action[1]        # pinned host memory the kernel reads directly
        st = _raw_stream(v.device.index)
        rc = v._sbr_step(v._h, *v._h_ptrs, st) or v._sbr_sync(v._h, st)   # two C calls: launch, wait
        if rc:
            _capi.check(rc, v._h)
        o = self._obs_row.tolist()
        state = self._state_row.copy()
        reward, done = float(self._reward_view[0]), bool(self._done_view[0])
        if not self._done:                        # a finished env ignores further calls (and leaves no trace record): keep the
            self._rewards.append(reward)          # per-call lists of trajectory() the same length as the records
            self._states.append(state)
            self._actions.append((float(action[0]), float(action[1])))
        self._done = done
        return (o[:9], o[9:]), state, reward, done, {}

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
        """The reference's sub-interval rows of the running episode.  The reference appends odeint's solution on an output grid
        of its own: the fill phase on linspace(0, T_fill, 252) (:296-313), every control interval on
        t_range = linspace(t, t + t_delta, int(((t + t_delta) - t)/dt)) (:1339, :1384: 9 or 10 points with the rounding of the
        span; t_range[1:] to t_t, x_out[1:] to x_t, x_out[:-1, k] to So_t / Ss_t / Sno_t / Snh_t, len - 1 copies of the set-points
        to u_DO_t / u_EC_t, :1359-1369, :876-892), and on the done call constant rows over the settle and draw grids and the idle
        phase on linspace(t_after_draw, t_cycle, n) (:1122-1155).  LSODA interpolates its own steps onto those grids; here the
        RK4 nodes of every span and the right-hand side at them come from the device (sbr_eval_substeps, replayed from the
        recorded start state, Kla and EC) and are interpolated by Hermite polynomials (quintic through three nodes, `_hermite`) -
        exact at the nodes, so the last row of an interval is the state step() returned.  The grids are formed with the
        reference's own NumPy calls: t_t equals the reference's list bit for bit."""
        from .. import _capi as K
        cfg, vec = self._vec.cfg, self._vec
        dt, t_delta, n = cfg.dt, cfg.t_delta, rec.shape[0]
        out = {k: [] for k in ("t_t", "x_t", "So_t", "Ss_t", "Sno_t", "Snh_t", "u_DO_t", "u_EC_t", "EC", "e_EC", "ie_EC", "dcv_EC")}
        conc = (("So_t", 8), ("Ss_t", 2), ("Sno_t", 9), ("Snh_t", 10))

        def np_(pair):
            return tuple(v.cpu().numpy() for v in pair)

        # ---- fill phase (reset): 252 rows, the first one the start state; Kla from the DO-PID at t = 0 (:1593-1617)
        x_pre, infl = self._x_prefill, self._influent
        rows_f = int((cfg.T_fill - 0) / dt)
        e0 = 0.0 - x_pre[8]
        kla_f = min(max(cfg.Kc_DO * e0 + (cfg.Kc_DO / cfg.tauI_DO) * 0.0, cfg.Kla_min), cfg.Kla_max)
        xs, dx = np_(vec.eval_substeps(x_pre[None], [kla_f], [cfg.T_fill / rows_f], loading=infl[None], kind=1, n_sub=rows_f))
        grid = np.linspace(0, cfg.T_fill, rows_f)
        rows = _hermite(xs[0], dx[0], cfg.T_fill, grid)
        rows[0], rows[-1] = xs[0, 0], xs[0, -1]
        out["t_t"] += grid.tolist()
        out["x_t"].append(rows)
        for name, j in conc:
            out[name] += rows[:-1, j].tolist()
        out["u_DO_t"] += [0, kla_f] * (rows_f // 2)                    # Kla * int(len(x_out)/len(Kla)) with Kla = [0, k_fill], :320
        out["u_EC_t"] += [0, 0.0] * (rows_f // 2)                      # EC likewise (:321)
        out["EC"] += [0, 0.0] * (rows_f // 2)                          # EC = [ec, 0] * 126 (:284, :1637, :324)
        out["e_EC"].append(0.0 - float(x_pre[9]))                      # Sim_filling's one entry: sp 0 - Sno[-1], Sno = [x_in[9]] (:1624)
        out["ie_EC"].append(0.0); out["dcv_EC"].append(0.0)            # t_start == 0 (:1630-1631); EC = 0 is inside its limits
        if n == 0:
            out["x_t"] = np.vstack(out["x_t"])
            return out

        # ---- control intervals
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
        xs1, dx1 = np_(vec.eval_substeps(x0[live], first_kla[live], span_of(t0[live]) / cfg.substeps, ec=first_ec[live]))
        two = np.nonzero(n_iv[live] >= 2)[0]
        if len(two):
            t_mid = t0[live][two] + t_delta
            xs2, dx2 = np_(vec.eval_substeps(xs1[two, -1], rec[live][two, K.TR_KLA], span_of(t_mid) / cfg.substeps,
                                             ec=rec[live][two, K.TR_EC]))

        def emit(t_start, nodes, slopes, u_do, u_ec, ec, pid, x_end=None):
            span = span_of(t_start)
            grid = np.linspace(t_start, t_start + t_delta, int(span / dt))          # the reference's t_range, bit for bit
            rows = _hermite(nodes, slopes, span, np.clip(grid - t_start, 0.0, span))
            # exact at the nodes; the last row of a call is the state step() returned (the replay closes V, Si, Xi and the
            # charge balance per substep, step() per interval: equal to rounding, pinned to the record)
            rows[0], rows[-1] = nodes[0], (nodes[-1] if x_end is None else x_end)
            out["t_t"] += grid[1:].tolist()
            out["x_t"].append(rows[1:])
            for name, j in conc:
                out[name] += rows[:-1, j].tolist()
            out["u_DO_t"] += [u_do] * (len(grid) - 1)
            out["u_EC_t"] += [u_ec] * (len(grid) - 1)
            out["EC"] += [ec] * (len(grid) - 1)                       # EC.append once (:1937 / :2025) + len(t_range) - 2 copies (:1957)
            out["e_EC"].append(pid[0]); out["ie_EC"].append(pid[1]); out["dcv_EC"].append(pid[2])     # one entry per INTERVAL

        second = {int(live[j]): i for i, j in enumerate(two)}
        x_last = None
        pid_first = rec[:, [K.TR_E_EC_FIRST, K.TR_IE_EC_FIRST, K.TR_DCV_EC_FIRST]]
        pid_last = rec[:, [K.TR_E_EC, K.TR_IE_EC, K.TR_DCV_EC]]
        terminal_call = (rec[:, K.TR_DONE] == 1.0) & bool(cfg.terminal)      # its record holds the state AFTER settle / draw / idle
        for i, k in enumerate(live):
            k = int(k)
            x_rec = None if terminal_call[k] else rec[k, K.TR_X0:K.TR_X0 + 14]
            if k in second:
                emit(t0[k], xs1[i], dx1[i], *setpoints(t0[k], self._actions[k]), first_ec[k], pid_first[k])
                j = second[k]
                emit(t0[k] + t_delta, xs2[j], dx2[j], rec[k, K.TR_U_DO], rec[k, K.TR_U_EC], rec[k, K.TR_EC], pid_last[k], x_rec)
                x_last = xs2[j, -1]
            else:
                emit(t0[k], xs1[i], dx1[i], rec[k, K.TR_U_DO], rec[k, K.TR_U_EC], rec[k, K.TR_EC], pid_last[k], x_rec)
                x_last = xs1[i, -1]

        # ---- the done call: settle and draw rows are constant (x before / after the draw, :2322-2323, :2411-2413), then idle
        if rec[-1, K.TR_DONE] == 1.0 and cfg.terminal and x_last is not None:
            t_last = rec[-1, K.TR_T]
            g_set = np.linspace(t_last, t_last + cfg.t_settle * cfg.t_cycle, int((cfg.t_settle * cfg.t_cycle) / t_delta))
            g_draw = np.linspace(g_set[-1], g_set[-1] + cfg.t_draw * cfg.t_cycle, int((cfg.t_draw * cfg.t_cycle) / t_delta))
            t_idle0 = g_draw[-1]
            g_idle = np.linspace(t_idle0, cfg.t_cycle, int((cfg.t_cycle - t_idle0) / dt))
            n_idle = len(g_idle)
            kla_idle = float(vec.ctrl_row(K.C_KLA_LAST)[0].item())       # Sim_idle's DO-PID output, left in Kla[-1] by the done call
            span = cfg.t_cycle - t_idle0
            xi, di = np_(vec.eval_substeps(x_last[None], [kla_idle], [span / n_idle], kind=3, n_sub=n_idle))
            r_idle = _hermite(xi[0], di[0], span, np.clip(g_idle - t_idle0, 0.0, span))
            r_idle[0], r_idle[-1] = xi[0, 0], xi[0, -1]
            x_out1 = np.vstack([np.repeat(x_last[None], len(g_set), 0), np.repeat(xi[0, :1], len(g_draw) - 1, 0)])
            t_all = np.concatenate([g_set, g_draw[1:], g_idle[1:]])
            out["t_t"] += t_all[1:].tolist()
            out["x_t"].append(np.vstack([x_out1, r_idle[1:]])[1:])
            for name, j in conc:
                out[name] += x_out1[:-1, j].tolist() + r_idle[:-1, j].tolist()
            out["u_DO_t"] += [out["u_DO_t"][-1]] * (len(t_all) - 1)                 # :1153-1154
            out["u_EC_t"] += [out["u_EC_t"][-1]] * (len(t_all) - 1)
            out["EC"] += [0] * (len(t_all) - 1)                                     # settle + draw (:2411-2412) and idle (:2593-2594)
        out["x_t"] = np.vstack(out["x_t"])
        return out

    def trajectory(self, as_dict=False, dense=False):
        """The reference's 18-tuple, in its order (gym_SBR_oneshot.py:1288):

            t_t, x_t, u_DO_t, u_EC_t, state_t, So_t, Ss_t, EC, Sno_t, dcv_EC, ie_EC, e_EC,
            reward_t, reward_EQI_t, reward_OCI_t, reward_AE_t, reward_EC_t, Snh_t

        Every list has ONE entry per step() call of the running episode, recorded on the device at the end of the call
        (for a call that runs two control intervals: of the second, like the reward).  The reference grows t_t, x_t, the
        four concentration lists, the two set-point lists and EC by the 8 or 9 rows of LSODA's output grid per interval, and
        its three PID lists (dcv_EC, ie_EC, e_EC) by one entry per INTERVAL, after one for the fill phase; the fixed-step
        integrator has no such grid, so those lists are sampled per call here - unless dense=True.  reward_t and the four reward
        diagnostics (module_reward_EQIOCI.py:109-112) are per call in the reference too.  state_t, which the reference leaves
        empty (its append is commented out, :436), holds the state vector step() returned.
        as_dict=True returns the same arrays by name (plus Kla, the DO controller's output).
        dense=True returns t_t, x_t, So_t, Ss_t, Sno_t, Snh_t, u_DO_t, u_EC_t and (round 4) EC, dcv_EC, ie_EC, e_EC as the
        reference grows them, entry for entry (`_dense_rows`): 252 rows of the fill phase, 8 or 9 rows per control interval, and
        after the done call the rows of settle, draw and idle (4767 time points and EC entries for a whole episode); 467 entries
        in each of the three PID lists - the fill phase's, then one per interval, two for a call that crosses a phase boundary.
        The last dense row of a call is the state step() returned (pinned to the record; the replayed nodes agree with it to
        rounding, ~1e-13 relative)."""
        from .. import _capi as K
        n = len(self._rewards)
        rec = self._trace[:n].cpu().numpy()[:, :, 0]
        rec = rec[np.isfinite(rec[:, K.TR_T])]          # (step() stops appending once the episode is done: every per-call list
        assert len(rec) == n == len(self._states)       # has one entry per recorded call)
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
