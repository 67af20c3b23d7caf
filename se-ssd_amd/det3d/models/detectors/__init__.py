from .voxelnet import VoxelNet  # noqa: F401
