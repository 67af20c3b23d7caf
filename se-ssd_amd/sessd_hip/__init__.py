"""MI355X-native SE-SSD hot path: Python face of libsessd_hip.so (device pointers come from torch)."""
from ._lib import lib, LIB_PATH, SIGNATURES, SessdError, check  # noqa: F401
from . import ops  # noqa: F401
