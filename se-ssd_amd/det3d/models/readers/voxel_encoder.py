"""mirrors det3d/models/readers/voxel_encoder.py:208-220 (VoxelFeatureExtractorV3)."""
import torch
from torch import nn

from sessd_hip import ops

from ..registry import READERS


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels):
        """features (M, max_points, ndim), num_voxels (M,) -> (M, num_input_features) mean over the filled slots."""
        return ops.vfe_mean(features.float().contiguous(), num_voxels.int().contiguous(), self.num_input_features)
