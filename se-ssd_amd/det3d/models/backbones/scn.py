"""mirrors det3d/models/backbones/scn.py:92-189 (SpMiddleFHD): same module tree => same state_dict keys
(`middle_conv.{0,3,...,39}.weight` in spconv's [kz,ky,kx,Cin,Cout] layout, BatchNorm1d at {1,4,...})."""
import numpy as np
import spconv
import torch
from spconv import SparseConv3d, SubMConv3d
from torch import nn

from ..registry import BACKBONES
from ..utils import build_norm_layer


@BACKBONES.register_module
class SpMiddleFHD(nn.Module):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleFHD", **kwargs):
        super().__init__()
        self.name = name
        self.chain_tables = True   # capacity mode: sites + neighbour tables of all layers from one ops.SparseChain run (spconv.ChainPlan)
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        bn = lambda c: build_norm_layer(norm_cfg, c)[1]
        self.middle_conv = spconv.SparseSequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="subm0"), bn(16), nn.ReLU(),
            SubMConv3d(16, 16, 3, bias=False, indice_key="subm0"), bn(16), nn.ReLU(),
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False), bn(32), nn.ReLU(),
            SubMConv3d(32, 32, 3, indice_key="subm1", bias=False), bn(32), nn.ReLU(),
            SubMConv3d(32, 32, 3, indice_key="subm1", bias=False), bn(32), nn.ReLU(),
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm2", bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm2", bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm2", bias=False), bn(64), nn.ReLU(),
            SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm3", bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm3", bias=False), bn(64), nn.ReLU(),
            SubMConv3d(64, 64, 3, indice_key="subm3", bias=False), bn(64), nn.ReLU(),
            SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=False), bn(64), nn.ReLU(),
        )

    def init_weights(self, pretrained=None):
        pass

    def __deepcopy__(self, memo):
        """A copy (the EMA teacher is a deep copy of the student) must not inherit the per-module chain plan -- its tables are
        keyed by id() of THIS module's layers, so a copied plan silently sends the copy down the slow per-layer path and shares
        no buffers safely -- nor the overflow flag buffer."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_plan", "_err_buf", "last_err"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def forward(self, voxel_features, coors, batch_size, input_shape, n_dev=None):
        """n_dev (ours; device int32[1]): the first n_dev[0] rows of voxel_features / coors are voxels, the tables are
        capacity-sized and no count is read back (spconv capacity mode: a capturable iteration). `self.last_err` then holds
        the device overflow flag of the pass."""
        sparse_shape = np.array([int(v) for v in input_shape[::-1]]) + [1, 0, 0]
        coors = coors.int()
        err = None
        if n_dev is not None:
            # the overflow flag of the pass: ONE buffer per module, allocated at the first (eager) capacity-mode call, so that it is
            # not a temporary inside a captured graph's private pool; cleared at the start of every pass
            if getattr(self, "_err_buf", None) is None or self._err_buf.device != coors.device:
                self._err_buf = torch.zeros((1,), dtype=torch.int32, device=coors.device)
            err = self._err_buf
            err.zero_()
        ret = spconv.SparseConvTensor(voxel_features, coors, sparse_shape, batch_size, n_dev=n_dev, err=err)
        self.last_err = ret.err
        if n_dev is not None and self.chain_tables:
            # every resolution's sites and every layer's neighbour table in one chain of ~8 launches (spconv.ChainPlan)
            key = spconv.ChainPlan.key_of(ret)
            if getattr(self, "_plan", None) is None or self._plan.key != key:
                self._plan = spconv.ChainPlan([m for m in self.middle_conv._modules.values() if isinstance(m, spconv.SparseConvolution)], ret)
            self._plan.run(ret)
        ret = self.middle_conv(ret)
        ret = ret.dense()
        N, C, D, H, W = ret.shape
        return ret.view(N, C * D, H, W)
