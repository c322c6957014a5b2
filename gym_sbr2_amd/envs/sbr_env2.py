"""`SbrEnv2` - the reference's per-cycle class (gym_SBR/envs/gym_SBR_env2.py:58, id `SBR-v2`) with the same surface,
backed by the batched HIP environment at N = 1:  reset() -> ndarray[3];  step(a[3]) -> (ndarray[3], float, True, {})."""
import numpy as np
import torch

from .. import _gymcompat as _gym
from ..cycle_env import SbrEnv2Vec
from .sbr_os import reference_randn


class SbrEnv2(_gym.Env):
    metadata = {"render.modes": ["human"]}

    def __init__(self, device=0, seed=None, rng=None):
        """rng: "numpy" - reset() draws the influent noise with np.random.randn(48) on the host, the reference's own draw
        (buffer_tank3.py:68, called from gym_SBR_env2.py:104), so `np.random.seed(k); env.reset()` gives the reference's cycle of
        seed k; "philox" - drawn on the device from `seed` + episode count.  Default: "numpy", or "philox" when a seed is given."""
        self.action_space = _gym.box([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])                   # :64
        self.observation_space = _gym.box([0.5, 0, 0], [1.33, 2.5, 2])                   # :66 (as declared upstream)
        self._vec = SbrEnv2Vec(1, device=device, out_dtype=torch.float64, action_dtype=torch.float64)
        if rng is None:
            rng = "philox" if seed is not None else "numpy"
        if rng not in ("numpy", "philox"):
            raise ValueError('rng must be "numpy" or "philox"')
        self._rng, self._seed, self._episodes, self.reward = rng, seed, 0, 0

    def seed(self, seed=None):
        if self._rng == "numpy":
            np.random.seed(seed)
        else:
            self._seed = seed
        return [seed]

    def reset(self, rnd=None, scenario=None, influent=None, carry_over=False):
        seed = (0 if self._seed is None else int(self._seed)) + self._episodes
        self._episodes += 1
        if rnd is None and influent is None and self._rng == "numpy":
            rnd = reference_randn(0 if scenario is None else int(scenario))      # scenario 0 (:104): one draw
        obs = self._vec.reset(seed=seed, scenario=None if scenario is None else [int(scenario)],
                              rnd=None if rnd is None else np.asarray(rnd, dtype=np.float64)[None],
                              influent=None if influent is None else np.asarray(influent, dtype=np.float64)[None],
                              carry_over=carry_over)
        return obs[0].cpu().numpy()

    def step(self, action):
        a = torch.tensor([[float(action[0]), float(action[1]), float(action[2])]], dtype=torch.float64)
        obs, reward, _ = self._vec.step(a)
        self.reward = float(reward[0].item())
        return obs[0].cpu().numpy(), self.reward, True, {}

    def diagnostics(self):
        names = ["Qw", "EQI", "OCI", "Ntot_eff", "COD_eff", "Snh_eff", "BOD5_eff", "Sno_eff", "Kla3_mean", "Kla5_mean",
                 "Kla8_mean", "Xf"]
        return dict(zip(names, self._vec.diag[0].tolist()))

    def render(self, mode="human", close=False):
        print("Reward for this episode: {}".format(self.reward))      # gym_SBR_env2.py:189-192

    def close(self):
        self._vec.close()
