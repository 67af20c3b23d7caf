"""numpy helpers of the predict path (mirrors det3d/core/bbox/box_np_ops.py:780-833,995-1004)."""
from sessd_hip.anchors import create_anchors_3d_range, get_valid_frustum, projection_matrix_to_CRT_kitti  # noqa: F401
