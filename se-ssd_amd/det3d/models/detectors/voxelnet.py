"""mirrors det3d/models/detectors/voxelnet_sessd.py:7-43 + single_stage.py:8-19 (VoxelNet / SingleStageDetector)."""
from torch import nn

from .. import builder
from ..registry import DETECTORS


@DETECTORS.register_module
class VoxelNet(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.bbox_head = builder.build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    def extract_feat(self, data):
        input_features = self.reader(data["voxels"], data["num_points_per_voxel"])
        x = self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"])
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward(self, example, is_ema=[False, None], return_loss=True, **kwargs):
        key_tag = "_raw" if is_ema[0] else ""
        voxels = example["voxels" + key_tag]
        coordinates = example["coordinates" + key_tag]
        num_points_per_voxel = example["num_points" + key_tag]
        num_voxels = example["num_voxels" + key_tag]
        batch_size = len(num_voxels)
        input_shape = example["shape" + key_tag][0]
        data = dict(voxels=voxels, num_points_per_voxel=num_points_per_voxel, coors=coordinates, batch_size=batch_size,
                    input_shape=input_shape)
        x = self.extract_feat(data)
        preds = self.bbox_head(x)
        if is_ema[0]:
            return preds
        if return_loss:
            return self.bbox_head.loss(example, preds, is_ema[1])
        return self.bbox_head.predict(example, preds, self.test_cfg)
