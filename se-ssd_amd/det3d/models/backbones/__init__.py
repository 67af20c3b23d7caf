from .scn import SpMiddleFHD  # noqa: F401
