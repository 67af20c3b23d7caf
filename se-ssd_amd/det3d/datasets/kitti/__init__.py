from .eval import get_official_eval_result  # noqa: F401
