from .eval import get_coco_eval_result, get_official_eval_result, get_official_eval_result_v2  # noqa: F401
