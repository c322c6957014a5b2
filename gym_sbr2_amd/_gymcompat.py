"""Which gym the reference-shaped classes plug into.

The reference's classes are `gym.Env` subclasses reached through `gym.envs.registration.register`
(gym_SBR/__init__.py:1-12, gym_SBR/envs/gym_SBR_oneshot.py:99-113).  `gym` is unpinned upstream (setup.py:3) and the
surface is the OLD gym API generation: `reset()` returns the observation only, there is no `seed=`/`options=` protocol,
`metadata` uses 'render.modes', and `SbrOS.step` returns its own 5-tuple `(obs, state, reward, done, {})`.  The
classes of this package speak exactly that generation, whichever library provides the base class:

  * `gym` importable         -> `Env = gym.Env`, `Box = gym.spaces.Box`              (what the reference imports)
  * else `gymnasium`         -> `Env = gymnasium.Env`, `Box = gymnasium.spaces.Box`  (base class and spaces only: gymnasium's
                                `make()` wrappers assume the NEW step/reset protocol, so the ids are registered there with the
                                passive checker and the order enforcer switched off)
  * neither (this image)     -> `Env = object`, `Box` = the small stand-in below; `gym_sbr2_amd.make()` still knows the ids
"""
import importlib

import numpy as np


class _Box:
    """Just enough of gym.spaces.Box when neither gym nor gymnasium is installed."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        self.shape, self.dtype = self.low.shape, dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains


def resolve():
    """(library name or None, Env base class, Box class) - first of gym, gymnasium that imports."""
    for name in ("gym", "gymnasium"):
        try:
            lib = importlib.import_module(name)
            spaces = importlib.import_module(name + ".spaces")
            return name, lib.Env, spaces.Box
        except Exception:          # not installed, or an installation that does not import: fall through to the next
            continue
    return None, object, _Box


LIBRARY, Env, Box = resolve()


def box(low, high, dtype=np.float32):
    """A Box over explicit bounds, with whichever Box class is in force."""
    return Box(np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype=dtype)
