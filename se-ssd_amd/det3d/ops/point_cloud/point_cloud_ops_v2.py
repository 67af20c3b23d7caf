"""mirrors det3d/ops/point_cloud/point_cloud_ops_v2.py:120-194 (points_to_voxel), on the MI355X.

numpy in -> numpy out like the reference (the data pipeline contract), torch device tensor in -> device tensors out.
Bit-exact with the reference's serial numba loop; unlike it, max_voxels is not limited to 65534 and any
range / voxel size works (the reference indexes a global (40,1600,1408) uint16 map)."""
import numpy as np
import torch

from sessd_hip import ops


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    if not reverse_index:
        raise NotImplementedError("only reverse_index=True (zyx coordinates) is on the SE-SSD path")
    as_numpy = isinstance(points, np.ndarray)
    dev = torch.device("cuda", torch.cuda.current_device()) if as_numpy else points.device
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev) if as_numpy else points.float().contiguous()
    r = ops.voxelize_batch([pts], [float(v) for v in voxel_size], [float(v) for v in coors_range], int(max_points),
                           int(max_voxels), with_batch_index=False, want_mean=False)
    m = int(r["prefix"][1].item())
    voxels, coors, num = r["voxels"][:m], r["coors"][:m], r["num_points"][:m]
    if as_numpy:
        return voxels.cpu().numpy(), coors.cpu().numpy(), num.cpu().numpy()
    return voxels, coors, num
