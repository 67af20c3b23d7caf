"""The data pipeline as a chain of stages (same contract as det3d/datasets/pipelines/compose.py): every stage maps
(res, info) -> (res, info); the chain is configured as a list whose items are either stage objects or the config's
dict(type="<registered stage>", ...). A stage that returns res = None drops the sample and ends the chain."""
from collections.abc import Sequence

from det3d.utils import build_from_cfg

from ..registry import PIPELINES


def _as_stage(item):
    if isinstance(item, dict):
        return build_from_cfg(item, PIPELINES)
    if callable(item):
        return item
    raise TypeError("transform must be callable or a dict")


@PIPELINES.register_module
class Compose(object):
    def __init__(self, transforms):
        if not isinstance(transforms, Sequence):
            raise AssertionError("transforms must be a sequence of stages / stage configs")
        self.transforms = [_as_stage(t) for t in transforms]

    def __call__(self, res, info):
        state = (res, info)
        for stage in self.transforms:
            state = stage(*state)
            if state[0] is None:
                return None
        return state

    def __repr__(self):
        body = "".join("\n    {0}".format(stage) for stage in self.transforms)
        return "{0}({1}\n)".format(type(self).__name__, body)
