"""Batched per-cycle environment `SBR-v2` on one MI355X: host-side mirror of the reference's SbrEnv2
(gym_SBR/envs/gym_SBR_env2.py:58).  One step() is a whole 12 h cycle (528 control intervals) in ONE kernel launch.
All numbers come from libsbr_amd.so; there is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .vec_env import SbrOSVec, _ptr


class SbrEnv2Vec(SbrOSVec):
    """N SBR-v2 environments.  reset() -> obs [N,3]; step(a [N,3] in [0,1]) -> obs [N,3], reward [N], done (all True).
    The three actions are the DO set-points (x 8 mg/L) of the two aerobic reaction phases and of the aerated idle phase."""

    def __init__(self, num_envs, **kw):
        super().__init__(num_envs, **kw)
        n, dev = self.num_envs, self.device
        self.cobs = torch.empty((n, _capi.NCYC_OBS), dtype=self.out_dtype, device=dev)
        self.diag = torch.empty((n, _capi.NCYC_DIAG), dtype=torch.float64, device=dev)
        self.all_done = torch.ones((n,), dtype=torch.uint8, device=dev)

    def reset(self, seed=0, scenario=None, rnd=None, influent=None, mask=None, carry_over=False):
        n = self.num_envs
        sc = self._dev(scenario, torch.int32, (n,))
        rn = self._dev(rnd, torch.float64, (n, _capi.NSAMP))
        inf = self._dev(influent, torch.float64, (n, _capi.NX))
        mk = self._dev(mask, torch.uint8, (n,))
        with torch.cuda.device(self.device):
            _capi.check(self.lib.sbr_cycle_reset(self._h, C.c_uint64(int(seed)), _ptr(sc), _ptr(rn), _ptr(inf), _ptr(mk),
                                                 1 if carry_over else 0, _ptr(self.cobs), self._stream()), self._h)
        self._keep = (sc, rn, inf, mk)
        return self.cobs

    def step(self, action, want_diag=True):
        a = action if (isinstance(action, torch.Tensor) and action.dtype == self.action_dtype and action.is_contiguous()
                       and action.device == self.device) else self._dev(action, self.action_dtype, (self.num_envs, 3))
        if tuple(a.shape) != (self.num_envs, 3):
            raise ValueError("action must have shape [N,3]")
        _capi.check(self.lib.sbr_cycle_step(self._h, _ptr(a), _ptr(self.cobs), _ptr(self.reward),
                                            _ptr(self.diag) if want_diag else None, self._stream()), self._h)
        self._keep_a = a
        return self.cobs, self.reward, self.all_done

    def rollout(self, *a, **k):
        raise NotImplementedError("the fused random-policy rollout belongs to SBROS-v1; SBR-v2 already runs a whole cycle per launch")
