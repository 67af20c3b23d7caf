from .registry import Registry, build_from_cfg  # noqa: F401
