"""mirrors det3d/models/necks/rpn_v1.py: SSFA (:119-235) and RPN (:23-116). Identical module trees (state_dict keys
`bottom_up_block_0.{1,2,4,5,7,8}`, `trans_0.{0,1}`, ...); forward runs the f32-MFMA conv kernels (eval mode)."""
import logging

import numpy as np
import torch
from torch import nn

from sessd_hip import ops
from sessd_hip.engine import fold_bn

from ..registry import NECKS
from ..utils import Sequential, build_norm_layer


class _Lowered:
    """conv(+BN+ReLU) blocks lowered to packed HIP launches, rebuilt when parameters change or move."""

    def __init__(self):
        self._cache = {}

    def __deepcopy__(self, memo):  # packed launches hold raw device pointers: a copied model re-packs on first use
        return _Lowered()

    def get(self, name, conv, bn, deconv=False):
        # every tensor that feeds the packing / BN folding, plus the raw-pointer write counter (ops.param_generation)
        tensors = [conv.weight, conv.bias] + ([] if bn is None else [bn.weight, bn.bias, bn.running_mean, bn.running_var])
        key = (conv.weight.data_ptr(), ops.param_generation()) + tuple(-1 if t is None else t._version for t in tensors)
        hit = self._cache.get(name)
        if hit is None or hit[0] != key:
            w = conv.weight.detach()
            pc = ops.pack_deconv2d_s2(w) if deconv else ops.pack_conv2d(w, conv.stride[0])
            if bn is not None:
                s, t = fold_bn(bn)
            else:
                s, t = None, (conv.bias.detach().float().contiguous() if conv.bias is not None else None)
            hit = (key, pc, s, t)
            self._cache[name] = hit
        return hit[1], hit[2], hit[3]


def _run_block(low, name, seq, x, residual=None):
    """seq = [pad?] (conv bn relu)*: run each conv with its folded BN and ReLU in the epilogue."""
    mods = list(seq._modules.values())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d):
            i += 1
            continue
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
            j = i + (2 if bn is not None else 1)
            relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
            pc, s, t = low.get("%s.%d" % (name, i), m, bn, isinstance(m, nn.ConvTranspose2d))
            last = (j + (1 if relu else 0)) >= len(mods)
            x = ops.conv2d(x, pc, s, t, relu, residual if last else None)
            i = j + (1 if relu else 0)
            continue
        raise TypeError("unexpected module in block: %r" % (m,))
    return x


@NECKS.register_module
class SSFA(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        self._layer_strides = ds_layer_strides
        self._num_filters = ds_num_filters
        self._layer_nums = layer_nums
        self._upsample_strides = us_layer_strides
        self._num_upsample_filters = us_num_filters
        self._num_input_features = num_input_features
        if norm_cfg is None:
            norm_cfg = dict(type="BN", eps=1e-3, momentum=0.01)
        self._norm_cfg = norm_cfg
        bn = lambda c: build_norm_layer(self._norm_cfg, c)[1]
        c3 = lambda i, o, s=1: nn.Conv2d(i, o, 3, stride=s, padding=1, bias=False)
        self.bottom_up_block_0 = Sequential(
            nn.ZeroPad2d(1), nn.Conv2d(128, 128, 3, stride=1, bias=False), bn(128), nn.ReLU(),
            c3(128, 128), bn(128), nn.ReLU(), c3(128, 128), bn(128), nn.ReLU())
        self.bottom_up_block_1 = Sequential(
            c3(128, 256, 2), bn(256), nn.ReLU(), c3(256, 256), bn(256), nn.ReLU(), c3(256, 256), bn(256), nn.ReLU())
        self.trans_0 = Sequential(nn.Conv2d(128, 128, 1, bias=False), bn(128), nn.ReLU())
        self.trans_1 = Sequential(nn.Conv2d(256, 256, 1, bias=False), bn(256), nn.ReLU())
        dc = lambda: nn.ConvTranspose2d(256, 128, 3, stride=2, padding=1, output_padding=1, bias=False)
        self.deconv_block_0 = Sequential(dc(), bn(128), nn.ReLU())
        self.deconv_block_1 = Sequential(dc(), bn(128), nn.ReLU())
        self.conv_0 = Sequential(c3(128, 128), bn(128), nn.ReLU())
        self.w_0 = Sequential(nn.Conv2d(128, 1, 1, bias=False), bn(1))
        self.conv_1 = Sequential(c3(128, 128), bn(128), nn.ReLU())
        self.w_1 = Sequential(nn.Conv2d(128, 1, 1, bias=False), bn(1))
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")
        self._low = _Lowered()
        self.fused_bn_train = True   # train mode: BatchNorm2d + ReLU on sessd_bn2d_relu_train_* (False: the torch modules)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def _forward_train(self, x):
        """Train mode (batch-statistics BatchNorm, autograd), composed as rpn_v1.py:220-235. The twelve conv layers that
        carry the FLOPs run forward AND backward on the HIP kernels (ops.Conv2dFunction), and so do the BatchNorm2d + ReLU that
        follow them (ops.bn2d_relu_train); the two 128->1 weight branches with their BatchNorm2d(1), the softmax and the blend are
        one fused pair of launches each way (ops.ssfa_fuse_train; the torch composition when a layer is not what it covers)."""
        def block(seq, inp):
            mods = [m for m in seq._modules.values() if not isinstance(m, nn.ZeroPad2d)]  # ZeroPad2d(1) + unpadded 3x3 == the
            i = 0                                                                          # padding-1 conv the kernels implement
            while i < len(mods):
                m = mods[i]
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    inp = ops.conv2d_module(inp, m)
                    i += 1
                elif isinstance(m, nn.BatchNorm2d) and type(m) is nn.BatchNorm2d and m.training and self.fused_bn_train:
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    inp = ops.bn2d_relu_train(inp, m, relu)     # batch statistics + normalise + ReLU: three launches
                    i += 2 if relu else 1
                else:
                    inp = m(inp)
                    i += 1
            return inp

        x_0 = block(self.bottom_up_block_0, x)
        x_1 = block(self.bottom_up_block_1, x_0)
        x_trans_0 = block(self.trans_0, x_0)
        x_trans_1 = block(self.trans_1, x_1)
        x_middle_0 = block(self.deconv_block_0, x_trans_1) + x_trans_0
        x_middle_1 = block(self.deconv_block_1, x_trans_1)
        x_output_0 = block(self.conv_0, x_middle_0)
        x_output_1 = block(self.conv_1, x_middle_1)
        def wbranch(seq, inp):  # Conv2d(C, 1, 1) (torch) + BatchNorm2d(1) without ReLU (the fused train-mode passes)
            y = seq[0](inp)
            bn = seq[1]
            if type(bn) is nn.BatchNorm2d and bn.training and self.fused_bn_train:
                return ops.bn2d_relu_train(y, bn, False)
            return bn(y)

        if self.fused_bn_train and ops.ssfa_fuse_train_covers(x_output_0, self.w_0[0], self.w_0[1], self.w_1[0], self.w_1[1]):
            # both weight branches, their BatchNorm2d(1), the softmax and the blend: two launches forward, two backward
            return ops.ssfa_fuse_train(x_output_0, x_output_1, self.w_0[0], self.w_0[1], self.w_1[0], self.w_1[1])
        w = torch.softmax(torch.cat([wbranch(self.w_0, x_output_0), wbranch(self.w_1, x_output_1)], dim=1), dim=1)
        return x_output_0 * w[:, 0:1] + x_output_1 * w[:, 1:]

    def forward(self, x):
        x = x.float().contiguous()
        if self.training:
            return self._forward_train(x)
        L = self._low
        x_0 = _run_block(L, "b0", self.bottom_up_block_0, x)
        x_1 = _run_block(L, "b1", self.bottom_up_block_1, x_0)
        x_trans_0 = _run_block(L, "t0", self.trans_0, x_0)
        x_trans_1 = _run_block(L, "t1", self.trans_1, x_1)
        x_middle_0 = _run_block(L, "d0", self.deconv_block_0, x_trans_1, residual=x_trans_0)
        x_middle_1 = _run_block(L, "d1", self.deconv_block_1, x_trans_1)
        x_output_0 = _run_block(L, "c0", self.conv_0, x_middle_0)
        x_output_1 = _run_block(L, "c1", self.conv_1, x_middle_1)
        s0, t0 = fold_bn(self.w_0[1])
        s1, t1 = fold_bn(self.w_1[1])
        w0 = self.w_0[0].weight.detach().reshape(-1).float().contiguous()
        w1 = self.w_1[0].weight.detach().reshape(-1).float().contiguous()
        return ops.ssfa_fuse(x_output_0, x_output_1, w0, w1, float(s0), float(t0), float(s1), float(t1))


@NECKS.register_module
class RPN(nn.Module):
    """SECOND-style RPN (rpn_v1.py:23-116): blocks of [pad, conv s, bn, relu, (conv, bn, relu)*n] + deconv/conv upsamplers,
    outputs concatenated. Only stride-1/2 3x3 convs and 1x1 / stride-2 3x3 transposed convs are lowered."""

    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        self._layer_strides, self._num_filters, self._layer_nums = ds_layer_strides, ds_num_filters, layer_nums
        self._upsample_strides, self._num_upsample_filters = us_layer_strides, us_num_filters
        self._num_input_features = num_input_features
        if norm_cfg is None:
            norm_cfg = dict(type="BN", eps=1e-3, momentum=0.01)
        self._norm_cfg = norm_cfg
        assert len(layer_nums) == len(ds_layer_strides) == len(ds_num_filters)
        assert len(us_num_filters) == len(us_layer_strides)
        self._upsample_start_idx = len(layer_nums) - len(us_layer_strides)
        in_filters = [num_input_features, *ds_num_filters[:-1]]
        blocks, deblocks = [], []
        for i, layer_num in enumerate(layer_nums):
            block = Sequential(nn.ZeroPad2d(1), nn.Conv2d(in_filters[i], ds_num_filters[i], 3, stride=ds_layer_strides[i], bias=False),
                               build_norm_layer(norm_cfg, ds_num_filters[i])[1], nn.ReLU())
            for j in range(layer_num):
                block.add(nn.Conv2d(ds_num_filters[i], ds_num_filters[i], 3, padding=1, bias=False))
                block.add(build_norm_layer(norm_cfg, ds_num_filters[i])[1])
                block.add(nn.ReLU())
            blocks.append(block)
            if i - self._upsample_start_idx >= 0:
                k = i - self._upsample_start_idx
                stride = us_layer_strides[k]
                if stride > 1:
                    up = nn.ConvTranspose2d(ds_num_filters[i], us_num_filters[k], stride, stride=stride, bias=False)
                else:
                    up = nn.Conv2d(ds_num_filters[i], us_num_filters[k], int(np.round(1 / stride)), stride=int(np.round(1 / stride)), bias=False)
                deblocks.append(Sequential(up, build_norm_layer(norm_cfg, us_num_filters[k])[1], nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")
        self._low = _Lowered()

    def forward(self, x):
        ups = []
        x = x.float().contiguous()
        for i in range(len(self.blocks)):
            x = _run_block(self._low, "blk%d" % i, self.blocks[i], x)
            if i - self._upsample_start_idx >= 0:
                de = self.deblocks[i - self._upsample_start_idx]
                if isinstance(de[0], nn.ConvTranspose2d):
                    ups.append(de(x))  # k == stride transposed conv: plain torch (not on the SE-SSD path)
                else:
                    ups.append(_run_block(self._low, "de%d" % i, de, x))
        return torch.cat(ups, dim=1) if len(ups) > 0 else x
