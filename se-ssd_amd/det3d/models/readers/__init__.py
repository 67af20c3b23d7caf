from .voxel_encoder import VoxelFeatureExtractorV3  # noqa: F401
