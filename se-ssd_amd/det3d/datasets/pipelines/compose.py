"""mirrors det3d/datasets/pipelines/compose.py: the pipeline as a chain of (res, info) -> (res, info) stages, built from the
config's list of dict(type=..., ...); a stage returning res = None drops the sample."""
import collections.abc

from det3d.utils import build_from_cfg

from ..registry import PIPELINES


@PIPELINES.register_module
class Compose(object):
    def __init__(self, transforms):
        assert isinstance(transforms, collections.abc.Sequence)
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                t = build_from_cfg(t, PIPELINES)
            elif not callable(t):
                raise TypeError("transform must be callable or a dict")
            self.transforms.append(t)

    def __call__(self, res, info):
        for t in self.transforms:
            res, info = t(res, info)
            if res is None:
                return None
        return res, info

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    %s" % t for t in self.transforms) + "\n)"
