from .sbr_env2 import SbrEnv2  # noqa: F401
from .sbr_os import SbrOS  # noqa: F401
