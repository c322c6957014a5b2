"""gym_sbr2_amd - MI355X-native batched SBR environment (hot path of SungKu/gym-SBR2: SBROS-v1).

Importing this package never touches the GPU.  `SbrOSVec` / `SbrOS` need a gfx950 device and
libsbr_amd.so (built in-tree by `__graft_entry__.build()`); there is no CPU fallback.
(The directory is `gym_sbr2_amd` because `gym-sbr2_amd` is not an importable Python name.)
"""
import os as _os

# Kernel arguments in device memory: with host-resident kernargs every k_step launch is 4 us slower (measured on MI355X,
# profiles/r01_notes.md).  It is the HIP runtime's default on this platform; pin it unless the user chose otherwise.  Only
# effective if set before the HIP runtime initialises, which importing this package early guarantees.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _capi  # noqa: F401,E402
from .registration import make, register_with_gym, registered_ids  # noqa: F401,E402

__version__ = "0.2.0"

# The reference registers its ids when `gym_SBR` is imported (gym_SBR/__init__.py:3-12); so does this package, with
# whichever of gym / gymnasium is installed (neither is in this image: then only gym_sbr2_amd.make() knows the ids).
# {library: [ids]}; a failed registration is a RuntimeWarning and is kept in registration.REGISTRATION_ERRORS, never dropped.
REGISTERED_WITH = register_with_gym()


def __getattr__(name):          # torch is imported only when the env classes are asked for
    if name == "SbrOSVec":
        from .vec_env import SbrOSVec
        return SbrOSVec
    if name == "SbrOS":
        from .envs import SbrOS
        return SbrOS
    if name == "SbrEnv2Vec":
        from .cycle_env import SbrEnv2Vec
        return SbrEnv2Vec
    if name == "SbrEnv2":
        from .envs import SbrEnv2
        return SbrEnv2
    if name == "ShardedSbrOS":
        from .sharding import ShardedSbrOS
        return ShardedSbrOS
    raise AttributeError(name)
