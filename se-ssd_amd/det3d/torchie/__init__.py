from .utils.config import Config, ConfigDict  # noqa: F401
