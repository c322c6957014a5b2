"""Opt-in import alias for code written against the reference package.

    import gym_sbr2_amd.compat as c; c.install_as_gym_SBR()
    import gym_SBR                      # -> this package: registers SBROS-v1 / SBR-v2 as gym_SBR/__init__.py:3-12 does
    from gym_SBR.envs import SbrOS      # -> gym_sbr2_amd.envs.SbrOS (gym_SBR/envs/__init__.py)
    np.random.seed(0); env = SbrOS(); env.reset()       # the reference's plant of seed 0: reset() draws np.random.randn(48)

Nothing is aliased unless install_as_gym_SBR() is called, and a `gym_SBR` that is already imported (the real reference) is
never replaced silently: that raises unless force=True."""
import sys
import types


def install_as_gym_SBR(force=False):
    if "gym_SBR" in sys.modules and not force and not getattr(sys.modules["gym_SBR"], "__sbr_amd_alias__", False):
        raise RuntimeError("a package named gym_SBR is already imported; pass force=True to shadow it for this process")
    import gym_sbr2_amd
    from gym_sbr2_amd import envs, registration
    pkg = types.ModuleType("gym_SBR")
    pkg.__doc__ = "alias of gym_sbr2_amd (MI355X-native SBROS-v1 / SBR-v2), installed by gym_sbr2_amd.compat"
    pkg.__sbr_amd_alias__ = True
    pkg.__path__ = []                        # a package, so that `import gym_SBR.envs` resolves through sys.modules
    pkg.envs = envs
    pkg.make, pkg.registered_ids = gym_sbr2_amd.make, gym_sbr2_amd.registered_ids
    sys.modules["gym_SBR"] = pkg
    sys.modules["gym_SBR.envs"] = envs
    # what importing the reference package does (gym_SBR/__init__.py:3-12).  `import gym_sbr2_amd` above has already registered
    # the ids: register_with_gym() skips ids that are in the library's registry, so an old-API gym does not see a duplicate
    pkg.REGISTERED_WITH = registration.register_with_gym()
    return pkg
