"""VoxelNet detector: reader -> sparse backbone -> BEV neck -> head (+ predict), the inference surface of
det3d/models/detectors/voxelnet_sessd.py:7-43 and single_stage.py:8-19. Same constructor kwargs and the same
`forward(example, is_ema=[False, None], return_loss=True)` contract:
  * `is_ema[0]` selects the un-augmented `*_raw` inputs (SE-SSD teacher pass) and returns raw head outputs,
  * `return_loss=False` returns `bbox_head.predict(...)`: a list of dict(box3d_lidar, scores, label_preds, metadata)."""
from torch import nn

from .. import builder
from ..registry import DETECTORS

_INPUT_KEYS = ("voxels", "coordinates", "num_points", "num_voxels", "shape")


@DETECTORS.register_module
class VoxelNet(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.bbox_head = builder.build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @property
    def with_neck(self):
        return getattr(self, "neck", None) is not None

    def extract_feat(self, data):
        feats = self.reader(data["voxels"], data["num_points_per_voxel"])
        if data.get("n_dev") is not None:  # capacity-sized inputs, live voxel count on the device (capturable iteration)
            bev = self.backbone(feats, data["coors"], data["batch_size"], data["input_shape"], n_dev=data["n_dev"])
        else:
            bev = self.backbone(feats, data["coors"], data["batch_size"], data["input_shape"])
        return self.neck(bev) if self.with_neck else bev

    def forward_preds(self, example, raw=False):
        """Head outputs of one pass: `raw` selects the un-augmented `*_raw` voxelization (teacher input)."""
        suffix = "_raw" if raw else ""
        voxels, coords, npts, nvox, shape = (example[k + suffix] for k in _INPUT_KEYS)
        # ours: `num_voxels_dev` (+ `_raw`), a device int32[1] with the batch's total voxel count, marks capacity-sized inputs
        bev = self.extract_feat(dict(voxels=voxels, num_points_per_voxel=npts, coors=coords, batch_size=len(nvox),
                                     input_shape=shape[0], n_dev=example.get("num_voxels_dev" + suffix)))
        return self.bbox_head(bev)

    def forward(self, example, is_ema=[False, None], return_loss=True, **kwargs):
        teacher_pass, teacher_preds = is_ema[0], is_ema[1]
        preds = self.forward_preds(example, raw=teacher_pass)
        if teacher_pass:
            return preds
        if return_loss:
            return self.bbox_head.loss(example, preds, teacher_preds)
        return self.bbox_head.predict(example, preds, self.test_cfg)
