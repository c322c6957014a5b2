from .sbr_os import SbrOS  # noqa: F401
