"""det3d/utils/config_tool.py:42-51 -- the one helper config.py imports from it (no protobuf dependency here)."""
from functools import reduce
from operator import mul


def get_downsample_factor(model_config):
    """BEV stride of the detector = prod(neck ds strides) / last neck us stride * backbone ds_factor (an int > 0)."""
    neck = model_config["neck"]
    factor = reduce(mul, neck.get("ds_layer_strides", [1]), 1)
    ups = neck.get("us_layer_strides", [])
    if len(ups):
        factor = factor / ups[-1]
    factor = int(factor * model_config["backbone"]["ds_factor"])
    if factor <= 0:
        raise AssertionError("downsample factor must be positive")
    return factor
