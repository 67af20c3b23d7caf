from .registry import BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, READERS  # noqa: F401
from .builder import (build_backbone, build_detector, build_head, build_loss, build_neck, build_reader)  # noqa: F401
from .readers import *  # noqa: F401,F403
from .backbones import *  # noqa: F401,F403
from .necks import *  # noqa: F401,F403
from .losses import *  # noqa: F401,F403
from .bbox_heads import *  # noqa: F401,F403
from .detectors import *  # noqa: F401,F403
