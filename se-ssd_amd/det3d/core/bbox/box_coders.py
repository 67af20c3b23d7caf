"""mirrors det3d/core/bbox/box_coders.py:60-106 (GroundBox3dCoder / GroundBox3dCoderTorch)."""
from det3d.core.bbox import box_torch_ops


class GroundBox3dCoderTorch(object):
    def __init__(self, linear_dim=False, vec_encode=False, n_dim=7, norm_velo=False):
        self.linear_dim = linear_dim
        self.vec_encode = vec_encode
        self.norm_velo = norm_velo
        self.n_dim = n_dim

    @property
    def code_size(self):
        return self.n_dim + 1 if self.vec_encode else self.n_dim

    def encode_torch(self, boxes, anchors):
        return box_torch_ops.second_box_encode(boxes, anchors, self.vec_encode, self.linear_dim)

    def decode_torch(self, boxes, anchors):
        return box_torch_ops.second_box_decode(boxes, anchors, self.vec_encode, self.linear_dim)
