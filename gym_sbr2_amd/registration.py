"""Environment ids.  The reference registers ten ids (gym_SBR/__init__.py:3-12); this build
implements the hot path `SBROS-v1` and the per-cycle `SBR-v2`.  The other eight raise a clear error (most of them cannot
run in the reference either).

The registered classes speak the reference's OLD gym API generation (`reset()` -> obs, `SbrOS.step()` -> the reference's own
5-tuple): see `_gymcompat`.  With `gym` the ids are registered exactly as upstream does (`register(id=..., entry_point=...)`); with
`gymnasium`, whose `make()` wraps an env in checkers that assume the NEW reset/step protocol, they are registered with
`disable_env_checker=True, order_enforce=False` so that `gymnasium.make('SBROS-v1')` hands out the class unwrapped."""
import importlib
import warnings

_REGISTRY = {"SBROS-v1": "gym_sbr2_amd.envs:SbrOS", "SBR-v2": "gym_sbr2_amd.envs:SbrEnv2"}
_NOT_BUILT = ["SBR-v0", "SBR-v1", "SBR-v4", "SBRCnt-v0", "SBRCnt-v1", "SBRCnt-v2", "SBRCntMA-v1", "SBROS-v2"]
REGISTRATION_ERRORS = {}          # {(library, env id): message} of the last register_with_gym() call


def registered_ids():
    return sorted(_REGISTRY)


def make(env_id, **kwargs):
    if env_id in _NOT_BUILT:
        raise NotImplementedError("%s is outside the paths this build accelerates (SBROS-v1, SBR-v2)" % env_id)
    if env_id not in _REGISTRY:
        raise KeyError("unknown environment id %r; available: %s" % (env_id, registered_ids()))
    mod, cls = _REGISTRY[env_id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)


def _register_one(reg, name, env_id, entry):
    if name == "gymnasium":
        try:
            reg.register(id=env_id, entry_point=entry, disable_env_checker=True, order_enforce=False)
            return
        except TypeError:          # a gymnasium/gym whose register() does not know these keywords
            pass
    reg.register(id=env_id, entry_point=entry)


def register_with_gym(strict=False):
    """Register the ids of this build with gym and/or gymnasium, whichever import (neither is in this image: then only
    gym_sbr2_amd.make() knows the ids).  Returns {library: [ids registered]}; a library that is not installed is simply absent.
    A registration that FAILS is never silent: strict=True raises, otherwise the failure is recorded in REGISTRATION_ERRORS
    and reported as a RuntimeWarning (an odd gym version must not make `import gym_sbr2_amd` unusable)."""
    done = {}
    REGISTRATION_ERRORS.clear()
    for name in ("gym", "gymnasium"):
        try:
            reg = importlib.import_module(name + ".envs.registration")
        except ImportError:
            continue
        done[name] = []
        for env_id, entry in sorted(_REGISTRY.items()):
            try:
                _register_one(reg, name, env_id, entry)
                done[name].append(env_id)
            except Exception as exc:          # noqa: BLE001 - reported below, never dropped
                REGISTRATION_ERRORS[(name, env_id)] = "%s: %s" % (type(exc).__name__, exc)
    if REGISTRATION_ERRORS:
        msg = "; ".join("%s.register(%r) failed: %s" % (k[0], k[1], v) for k, v in sorted(REGISTRATION_ERRORS.items()))
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return done
