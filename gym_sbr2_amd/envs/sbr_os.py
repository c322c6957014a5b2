"""`SbrOS` - the reference's single-environment class (gym_SBR/envs/gym_SBR_oneshot.py:99,
id `SBROS-v1`) with the same surface, backed by the batched HIP environment at N = 1.

    reset()                      -> (obs_DO: list[9], obs_EC: list[9])                     (:438)
    step([u_DO, u_EC])           -> (obs, state: ndarray[15], reward: float, done: bool, {})  (:1273)
    get_available_actions(...)   -> [ndarray, ndarray]                                     (:440-459)

Outputs are float64 like the reference's.  Differences, all deliberate: state lives on the GPU
instead of module globals (so several instances can coexist), the influent noise comes from an
explicit `seed`/`rnd` instead of the global numpy RNG, and the 18 lists of `trajectory()` are
sampled once per step() call (see its docstring).
"""
import numpy as np
import torch

from ..vec_env import SbrOSVec


class _Box:
    """Just enough of gym.spaces.Box when neither gym nor gymnasium is installed."""

    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        self.shape, self.dtype = self.low.shape, dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class SbrOS:
    metadata = {"render.modes": ["human"]}                      # :101

    def __init__(self, device=0, seed=None, reward=None):
        # the reference declares stale spaces (:106-113); these are the real ones of step()
        self.action_space = _Box([0.0, 0.0], [8.0, 15.0])
        self.observation_space = _Box(np.full(18, -np.inf), np.full(18, np.inf))
        # reward: None / "eqi_oci" = the reference's (module_reward_EQIOCI.py); "g2anet", "oci" = the other reward modules
        self._vec = SbrOSVec(1, device=device, out_dtype=torch.float64, action_dtype=torch.float64, reward=reward)
        self._vec.enable_host_io()        # the kernel reads the action from, and writes its outputs to, pinned host memory
        self._seed = seed
        self._episodes = 0
        self._rewards, self._states = [], []

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
        self._rewards, self._states = [], []
        self._trace = self._vec.enable_trace(1, 464)
        obs = self._vec.reset(seed=seed, carry_over=carry_over,
                              scenario=None if scenario is None else [int(scenario)],
                              rnd=None if rnd is None else np.asarray(rnd, dtype=np.float64)[None],
                              influent=None if influent is None else np.asarray(influent, dtype=np.float64)[None])
        return self._split(obs[0].cpu())

    def step(self, action):
        obs, state, reward, done = self._vec.step_host([[float(action[0]), float(action[1])]])
        state = state[0].copy()
        reward, done = float(reward[0]), bool(done[0])
        self._rewards.append(reward)
        self._states.append(state)
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

    def trajectory(self, as_dict=False):
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
        as_dict=True returns the same arrays by name (plus Kla, the DO controller's output)."""
        from .. import _capi as K
        n = len(self._rewards)
        rec = self._trace[:n].cpu().numpy()[:, :, 0]
        x_t = rec[:, K.TR_X0:K.TR_X0 + 14]
        cols = {"t_t": rec[:, K.TR_T], "x_t": x_t, "u_DO_t": rec[:, K.TR_U_DO], "u_EC_t": rec[:, K.TR_U_EC],
                "state_t": [s.copy() for s in self._states], "So_t": x_t[:, 8], "Ss_t": x_t[:, 2], "EC": rec[:, K.TR_EC],
                "Sno_t": x_t[:, 9], "dcv_EC": rec[:, K.TR_DCV_EC], "ie_EC": rec[:, K.TR_IE_EC], "e_EC": rec[:, K.TR_E_EC],
                "reward_t": rec[:, K.TR_REWARD], "reward_EQI_t": rec[:, K.TR_R_EQI], "reward_OCI_t": rec[:, K.TR_R_OCI],
                "reward_AE_t": rec[:, K.TR_R_AE], "reward_EC_t": rec[:, K.TR_R_EC], "Snh_t": x_t[:, 10]}
        if as_dict:
            cols["Kla"] = rec[:, K.TR_KLA]
            return cols
        order = ("t_t x_t u_DO_t u_EC_t state_t So_t Ss_t EC Sno_t dcv_EC ie_EC e_EC reward_t reward_EQI_t reward_OCI_t "
                 "reward_AE_t reward_EC_t Snh_t").split()
        return tuple(cols[k] if k in ("x_t", "state_t") else cols[k].tolist() for k in order)

    def render(self, mode="human"):
        return None

    def close(self):
        self._vec.close()
