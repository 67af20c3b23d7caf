from .mg_head_sessd import Head, MultiGroupHead  # noqa: F401
