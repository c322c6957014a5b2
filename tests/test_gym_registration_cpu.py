"""The gym side of the drop-in boundary, on CPU: neither gym nor gymnasium is in this image, so a stand-in
module plays the library - the same trick oracle/gen_golden.py uses to import the reference.  Checked: the env classes derive
from the library's Env (gym_SBR_oneshot.py:99), their spaces are the library's Box, the ids the reference registers for these
classes (gym_SBR/__init__.py:5,11) land in the library's registry with resolvable entry points, and a registration that fails
is never silent."""
import importlib
import sys
import types
import warnings

import numpy as np
import pytest


def _stand_in(name, fail_on=None, new_keywords=False):
    lib = types.ModuleType(name)

    class Env:
        metadata = {}

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = np.asarray(low), np.asarray(high), np.asarray(low).shape, dtype

    spaces, envs, registration = types.ModuleType(name + ".spaces"), types.ModuleType(name + ".envs"), types.ModuleType(name + ".envs.registration")
    registry = {}

    def register(id, entry_point=None, **kw):      # noqa: A002
        if kw and not new_keywords:
            raise TypeError("register() got an unexpected keyword argument %r" % sorted(kw)[0])
        if id == fail_on:
            raise ValueError("id %s refused" % id)
        if id in registry:                        # what an old-API gym does (gym.error.Error: Cannot re-register id)
            raise RuntimeError("Cannot re-register id: %s" % id)
        registry[id] = dict(entry_point=entry_point, **kw)

    spaces.Box, registration.register, registration.registry = Box, register, registry
    envs.registration = registration
    lib.Env, lib.spaces, lib.envs = Env, spaces, envs
    return {name: lib, name + ".spaces": spaces, name + ".envs": envs, name + ".envs.registration": registration}


@pytest.fixture
def fresh_package(monkeypatch):
    """Import gym_sbr2_amd's gym-facing modules afresh under whatever stand-ins the test installed, and restore afterwards."""
    mods = ["gym_sbr2_amd._gymcompat", "gym_sbr2_amd.envs.sbr_os", "gym_sbr2_amd.envs.sbr_env2", "gym_sbr2_amd.envs"]
    saved = {m: sys.modules.get(m) for m in mods}

    def load(stand_ins):
        for k, v in stand_ins.items():
            monkeypatch.setitem(sys.modules, k, v)
        for m in mods:
            sys.modules.pop(m, None)
        return importlib.import_module("gym_sbr2_amd._gymcompat"), importlib.import_module("gym_sbr2_amd.envs")
    yield load
    for m in mods:
        sys.modules.pop(m, None)
        if saved[m] is not None:
            sys.modules[m] = saved[m]


def test_without_gym_the_classes_are_plain_and_only_make_knows_the_ids(fresh_package):
    assert "gym" not in sys.modules and "gymnasium" not in sys.modules      # the image has neither
    compat, envs = fresh_package({})
    import gym_sbr2_amd
    assert compat.LIBRARY is None and compat.Env is object
    assert gym_sbr2_amd.registration.register_with_gym() == {}
    assert gym_sbr2_amd.registered_ids() == ["SBR-v2", "SBROS-v1"]
    with pytest.raises(NotImplementedError):
        gym_sbr2_amd.make("SBR-v0")
    b = compat.box([0.0, 0.0], [8.0, 15.0])
    assert b.contains(np.array([1.0, 2.0], dtype=np.float32)) and not b.contains(np.array([9.0, 2.0]))


@pytest.mark.parametrize("name", ["gym", "gymnasium"])
def test_classes_derive_from_the_library_env_and_register_like_the_reference(fresh_package, name):
    mods = _stand_in(name, new_keywords=(name == "gymnasium"))
    compat, envs = fresh_package(mods)
    lib = mods[name]
    assert compat.LIBRARY == name and compat.Env is lib.Env and compat.Box is lib.spaces.Box
    assert issubclass(envs.SbrOS, lib.Env) and issubclass(envs.SbrEnv2, lib.Env)          # gym_SBR_oneshot.py:99, gym_SBR_env2.py:58
    assert envs.SbrOS.metadata == {"render.modes": ["human"]}                            # :101
    from gym_sbr2_amd import registration
    done = registration.register_with_gym(strict=True)
    assert done == {name: ["SBR-v2", "SBROS-v1"]} and not registration.REGISTRATION_ERRORS
    reg = mods[name + ".envs.registration"].registry
    assert set(reg) == {"SBROS-v1", "SBR-v2"}                                             # gym_SBR/__init__.py:5, :11
    for env_id, cls in (("SBROS-v1", envs.SbrOS), ("SBR-v2", envs.SbrEnv2)):
        mod, attr = reg[env_id]["entry_point"].split(":")                                 # what gym.make() would resolve
        assert getattr(importlib.import_module(mod), attr) is cls
    if name == "gymnasium":       # new-protocol checkers off: the classes speak the reference's old API generation
        assert reg["SBROS-v1"]["disable_env_checker"] is True and reg["SBROS-v1"]["order_enforce"] is False
    assert isinstance(compat.box([0, 0], [8, 15]), lib.spaces.Box)
    # a second call finds the ids in the registry and does not register them again (the stand-in raises on a duplicate)
    assert registration.register_with_gym(strict=True) == {name: ["SBR-v2", "SBROS-v1"]}


def test_only_the_library_the_classes_derive_from_is_registered_with(fresh_package):
    """With both libraries installed the classes subclass gym.Env (gym is tried first): registering them with gymnasium too
    would hand gymnasium.make() a class it refuses."""
    mods = dict(_stand_in("gym"))
    mods.update(_stand_in("gymnasium", new_keywords=True))
    compat, envs = fresh_package(mods)
    from gym_sbr2_amd import registration
    assert compat.LIBRARY == "gym" and issubclass(envs.SbrOS, mods["gym"].Env)
    assert registration.register_with_gym(strict=True) == {"gym": ["SBR-v2", "SBROS-v1"]}
    assert set(mods["gym.envs.registration"].registry) == {"SBROS-v1", "SBR-v2"} and not mods["gymnasium.envs.registration"].registry


def test_a_registration_module_that_does_not_import_is_reported(fresh_package, monkeypatch):
    """Not only ImportError: an old gym under a new numpy dies with AttributeError inside gym.envs.registration; `import
    gym_sbr2_amd` must survive that and say so."""
    mods = _stand_in("gym")
    fresh_package(mods)
    from gym_sbr2_amd import registration
    real_import = importlib.import_module

    def broken(name, *a, **k):
        if name == "gym.envs.registration":
            raise AttributeError("module 'numpy' has no attribute 'bool8'")
        return real_import(name, *a, **k)
    monkeypatch.setattr(registration.importlib, "import_module", broken)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert registration.register_with_gym() == {}
    assert ("gym", "*") in registration.REGISTRATION_ERRORS and "bool8" in registration.REGISTRATION_ERRORS[("gym", "*")]
    assert any(issubclass(x.category, RuntimeWarning) for x in w)


def test_a_failed_registration_is_reported_not_swallowed(fresh_package):
    fresh_package(_stand_in("gym", fail_on="SBROS-v1"))
    from gym_sbr2_amd import registration
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        done = registration.register_with_gym()
    assert done == {"gym": ["SBR-v2"]}                      # says which ids made it, library by library
    assert ("gym", "SBROS-v1") in registration.REGISTRATION_ERRORS and "refused" in registration.REGISTRATION_ERRORS[("gym", "SBROS-v1")]
    assert any(issubclass(x.category, RuntimeWarning) and "SBROS-v1" in str(x.message) for x in w)
    with pytest.raises(RuntimeError, match="SBROS-v1"):
        registration.register_with_gym(strict=True)


def test_opt_in_alias_for_code_written_against_the_reference_package(fresh_package, monkeypatch):
    """`import gym_SBR` / `from gym_SBR.envs import SbrOS` (gym_SBR/__init__.py, gym_SBR/envs/__init__.py) resolve to this package
    after gym_sbr2_amd.compat.install_as_gym_SBR() - opt-in, and never over a gym_SBR that is already imported."""
    mods = _stand_in("gym")
    compat_mod, envs = fresh_package(mods)
    for name in ("gym_SBR", "gym_SBR.envs"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    from gym_sbr2_amd import compat
    pkg = compat.install_as_gym_SBR()
    import gym_SBR
    from gym_SBR.envs import SbrEnv2, SbrOS
    assert gym_SBR is pkg and SbrOS is envs.SbrOS and SbrEnv2 is envs.SbrEnv2 and gym_SBR.REGISTERED_WITH == {"gym": ["SBR-v2", "SBROS-v1"]}
    assert set(mods["gym.envs.registration"].registry) == {"SBROS-v1", "SBR-v2"}
    assert compat.install_as_gym_SBR() is not None                       # re-installing the alias over itself is fine
    monkeypatch.setitem(sys.modules, "gym_SBR", types.ModuleType("gym_SBR"))      # ... but not over somebody else's gym_SBR
    with pytest.raises(RuntimeError):
        compat.install_as_gym_SBR()
    assert compat.install_as_gym_SBR(force=True).__sbr_amd_alias__
    for name in ("gym_SBR", "gym_SBR.envs"):
        monkeypatch.delitem(sys.modules, name, raising=False)
