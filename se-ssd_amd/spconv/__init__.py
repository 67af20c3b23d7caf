"""spconv-v1 compatible module surface (only what det3d/models/backbones/scn.py touches), running on the
gfx950 HIP kernels of libsessd_hip.so. Mirrors: spconv.SparseConvTensor, SubMConv3d, SparseConv3d,
SparseSequential, SparseModule (reference call sites scn.py:4,9,24-44,46,106-148,182-184).

This generic path keeps the spconv API contract (`.features` has exactly N rows), which costs one
device->host read of the site count after every strided conv. The fused engine (sessd_hip.engine) does not.

CAPACITY MODE (SparseConvTensor(..., n_dev=count)): the tables keep a fixed row CAPACITY and the live row count stays on the
device (`n_dev`, int32[1]); a strided conv sizes its output by `capacity_growth` instead of reading the count back, every
kernel takes the device count (rows beyond it are never read), and nothing synchronises or copies from the host -- which is
what lets a whole training iteration (sessd_hip.train.TrainStep.capture) be ONE captured graph. Overflow of a capacity is
reported in `err` (device int, checked by the caller after the step)."""

# capacity of a strided conv's output level relative to its input level, in the order of the strided convs of SpMiddleFHD.
# Observed ratios on the un-augmented synthetic scans: 0.96 - 1.28, 0.5, 0.4, 0.85 (the inference engine sizes level 1 at 1.5).
# TRAINING sees the student's AUGMENTED cloud (rotation +-pi/4, scaling): rotated scan lines no longer run along the voxel rows
# and the first strided conv dilates them into many more output sites -- per frame 1.35 on average, 2.0 at worst, per batch of four
# up to 1.65 .. 1.9 (scripts/r6_sparse_overflow_scan.py: with 1.5 one student batch in ten overflowed level 1, silently until round
# 6's sticky flag). Level 1 is therefore sized at 2.5 x the voxel capacity; the deeper factors keep the absolute capacities of
# rounds 1 - 5 (2.5 x 0.6 = 1.5 x 1.0).
CAPACITY_GROWTH = (2.5, 0.6, 0.75, 0.75)
import math

import numpy as np
import torch
from torch import nn

from sessd_hip import ops


def _t3(v):
    return [int(v)] * 3 if isinstance(v, (int, np.integer)) else [int(x) for x in v]


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, n_dev=None, err=None, growth=None):
        self.features = features
        self.indices = indices.int().contiguous()
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self.n_dev = n_dev          # capacity mode: live rows (device int32[1]); None: every row is a site
        self.err = err              # capacity mode: device int32[1], non-zero after a capacity overflow
        self.growth = list(CAPACITY_GROWTH if growth is None else growth)  # consumed front to back by the strided convs
        if n_dev is not None and err is None:
            self.err = torch.zeros((1,), dtype=torch.int32, device=self.indices.device)

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first=True):
        out = ops.sparse_to_dense(self.features, self.indices, self.spatial_shape, self.batch_size, self.n_dev)  # (B,C,D,H,W), HIP scatter
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    def _hash(self):
        h = self.indice_dict.get("__hash__")
        if h is None:
            n = self.n_dev if self.n_dev is not None else torch.tensor([self.indices.shape[0]], dtype=torch.int32,
                                                                       device=self.indices.device)
            h = (ops.sparse_hash_build(self.indices, n, self.spatial_shape), n)
            self.indice_dict["__hash__"] = h
        return h


class SparseModule(nn.Module):
    pass


class ChainPlan:
    """Capacity mode: the site tables of every deeper resolution and the neighbour tables of EVERY conv layer of a sequence, built
    together by ops.SparseChain (csrc/sparse_sites.hip: ~8 launches for the 14 layers of SpMiddleFHD) instead of 3 - 5 launches
    per resolution (hash build, site generation, one rulebook per indice_key; 50 launches and their table fills for the two
    networks of a training iteration). Buffers are allocated once per plan and reused by every pass; deeper levels are numbered
    in ascending (b, z, y, x) order. A conv layer finds its tables through the tensor's indice_dict["__chain__"]."""

    def __init__(self, convs, x):
        steps, caps, jobs, subm_job = [], [], [], {}
        self.job_of, self.level_of = {}, {}
        level, shape, growth, cap = 0, list(x.spatial_shape), list(x.growth), int(x.indices.shape[0])
        for m in convs:
            if m.subm:
                key = (level, tuple(m.kernel_size))
                if key not in subm_job:
                    subm_job[key] = len(jobs)
                    jobs.append((level, level, m.kernel_size, 1, [k // 2 for k in m.kernel_size]))
                self.job_of[id(m)] = subm_job[key]
            else:
                shape = [(d + 2 * p - k) // s + 1 for d, k, s, p in zip(shape, m.kernel_size, m.stride, m.padding)]
                g = growth.pop(0) if growth else float(min(int(np.prod(m.kernel_size)), 8))
                cap = max(64, (min(x.batch_size * int(np.prod(shape)), int(math.ceil(g * cap))) + 63) // 64 * 64)
                steps.append((m.kernel_size, m.stride, m.padding))
                caps.append(cap)
                self.job_of[id(m)] = len(jobs)
                jobs.append((level, level + 1, m.kernel_size, m.stride, m.padding))
                level += 1
                self.level_of[id(m)] = level
        self.key = ChainPlan.key_of(x)
        self.chain = ops.SparseChain(x.spatial_shape, steps, caps, x.batch_size, jobs, x.indices.device)
        self.chain.bind_tables(int(x.indices.shape[0]))
        # The tables are allocated ONCE and rewritten by every run() through raw pointers (no autograd version bump), while
        # IndiceConvFunction saves them for backward: a second forward of the same backbone before the first one's backward
        # (gradient accumulation over two views) would silently differentiate the first pass with the second pass's tables.
        # Every run() bumps `generation`; a backward whose tables were overwritten raises (round-3 advisor finding).
        self.generation = 0

    @staticmethod
    def key_of(x):
        return (int(x.indices.shape[0]), tuple(x.spatial_shape), x.batch_size, str(x.indices.device), tuple(x.growth))

    def run(self, x):
        h, n = x._hash()
        self.chain.run(x.indices, n.data_ptr(), int(x.indices.shape[0]), h, x.err, clear=True)
        self.generation += 1
        x.indice_dict["__chain__"] = self

    def tables(self, conv):
        j = self.job_of[id(conv)]
        return self.chain.nbr[j], self.chain.tile_mask[j]

    def level(self, conv):
        l = self.level_of[id(conv)]
        return self.chain.indices[l - 1], self.chain.n_dev[l - 1], self.chain.shapes[l]


def _adjoint_pack(weight, reverse_offsets):
    """Packed weight of the conv that computes a sparse layer's data gradient (per offset W_k^T; offsets reversed for a
    submanifold layer): one launch from the stored weight where the pack kernel covers the shape, else flip / transpose / pack."""
    cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
    steps = cout // 4
    if weight.is_cuda and cin % 16 == 0 and cout % 4 == 0 and (steps <= 4 or steps % 4 == 0):
        return ops.packed_sparse_weight(weight, "adj_rev" if reverse_offsets else "adj")
    w = weight.detach().reshape(-1, cin, cout)
    if reverse_offsets:
        w = w.flip(0)
    return ops.sparse_pack_weight(w.transpose(1, 2).contiguous())


class IndiceConvFunction(torch.autograd.Function):
    """Differentiable sparse convolution y[j] = sum_k W_k^T x[nbr[k][j]] (+ bias) on the HIP kernels.
    backward: dx through the transposed rulebook with the same output-stationary kernel, dW by the site-reduction
    GEMM kernel (sessd_sparse_conv_wgrad), db = column sums. Deterministic (no atomics on floats)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, nbr, tm, n_out, n_in=None, subm=False, guard=None):
        """guard: (ChainPlan, generation) when nbr / tm are the plan's shared buffers -- checked again in backward."""
        ctx.guard = guard
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        out = ops.sparse_conv(feats, nbr, tm, n_out, ops.packed_sparse_weight(weight, "fwd"), cin, cout, None, bias, relu=False)
        if n_in is None:
            ctx.save_for_backward(feats, weight, nbr, tm, n_out)
        else:  # capacity mode: the input table's live row count lives on the device
            ctx.save_for_backward(feats, weight, nbr, tm, n_out, n_in)
        ctx.has_bias = bias is not None
        ctx.subm = bool(subm)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight, nbr, tm, n_out = ctx.saved_tensors[:5]
        if ctx.guard is not None and ctx.guard[0].generation != ctx.guard[1]:
            raise RuntimeError("the neighbour tables of this pass were overwritten by a later forward of the same backbone "
                               "(spconv.ChainPlan reuses one set of buffers): run backward before the next forward, or set "
                               "SpMiddleFHD.chain_tables = False for passes that must stay alive")
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        kv = nbr.shape[0]
        g = grad_out.float().contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and ctx.subm:
            # Submanifold conv: input and output sites are the same set and the offsets are centrally symmetric
            # (delta[K-1-k] = -delta[k]), so nbr[k][j] = i  <=>  nbr[K-1-k][i] = j: the transposed rulebook IS the forward table
            # read with the offsets reversed. dx[i] = sum_k W_k dy[nbrT[k][i]] = sum_k' W_{K-1-k'} dy[nbr[k'][i]] -- the forward
            # kernel on the forward tables with the per-offset weights reversed and transposed; no transpose launch, no fills
            # (10 of the 14 layers of SpMiddleFHD: 0.4 ms of rulebook work per iteration).
            gx = ops.sparse_conv(g, nbr, tm, n_out, _adjoint_pack(weight, True), cout, cin, None, None, relu=False)
        elif ctx.needs_input_grad[0]:
            n_in = feats.shape[0]
            nbr_t, tm_t = ops.sparse_rulebook_transpose(nbr, n_out, n_in)
            n_in_dev = ctx.saved_tensors[5] if len(ctx.saved_tensors) > 5 else torch.tensor([n_in], dtype=torch.int32,
                                                                                              device=feats.device)
            gx = ops.sparse_conv(g, nbr_t, tm_t, n_in_dev, _adjoint_pack(weight, False), cout, cin, None, None, relu=False)
        if ctx.needs_input_grad[1]:
            gw = ops.sparse_conv_wgrad(feats, g, nbr, tm, n_out, cin, cout).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb, None, None, None, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse and _t3(dilation) == [1, 1, 1]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _t3(kernel_size), _t3(stride), _t3(padding)
        self.subm, self.indice_key = subm, indice_key
        # spconv v1 weight layout: [kz, ky, kx, Cin, Cout]
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None

    def reset_parameters(self):
        fan_in = self.in_channels * int(np.prod(self.kernel_size))
        bound = math.sqrt(6.0 / ((1 + 5) * fan_in))  # kaiming_uniform(a=sqrt(5))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)

    def _wpk(self):
        if ops._registry_for(self.weight) is not None:   # inside a training iteration: kept and re-packed with all other layers
            return ops.packed_sparse_weight(self.weight, "fwd")
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device), ops.param_generation())
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.sparse_pack_weight(self.weight))
        return self._packed[1]

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        plan = x.indice_dict.get("__chain__") if x.n_dev is not None else None
        if plan is not None and id(self) not in plan.job_of:
            plan = None
        if plan is None:
            in_hash, n_in = x._hash()
        if plan is not None:   # sites and tables of this layer were built with all the others (ChainPlan): no hash of this level
            n_in = x.n_dev
            nbr, tm = plan.tables(self)
            if self.subm:
                out = SparseConvTensor(None, x.indices, x.spatial_shape, x.batch_size, n_dev=x.n_dev, err=x.err, growth=x.growth)
                out.indice_dict = x.indice_dict
                n_out = n_in
            else:
                oidx, n_out, oshape = plan.level(self)
                out = SparseConvTensor(None, oidx, oshape, x.batch_size, n_dev=n_out, err=x.err, growth=x.growth[1:])
                out.indice_dict["__chain__"] = plan
        elif self.subm:
            cached = x.find_indice_pair(self.indice_key)
            if cached is None:
                nbr, tm = ops.sparse_rulebook(x.indices, n_in, self.kernel_size, 1, [k // 2 for k in self.kernel_size], in_hash)
                cached = (nbr, tm)
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = cached
            nbr, tm = cached
            out = SparseConvTensor(None, x.indices, x.spatial_shape, x.batch_size, n_dev=x.n_dev, err=x.err, growth=x.growth)
            out.indice_dict = x.indice_dict
            n_out = n_in
        elif x.n_dev is not None:
            # capacity mode: the output level keeps `growth` x the input capacity (at most every cell), its row count stays on
            # the device; nothing is read back
            oshape = [(d + 2 * p - k) // s + 1 for d, k, s, p in zip(x.spatial_shape, self.kernel_size, self.stride, self.padding)]
            cells = x.batch_size * int(np.prod(oshape))
            g = x.growth[0] if x.growth else float(min(int(np.prod(self.kernel_size)), 8))
            cap = max(64, (min(cells, int(math.ceil(g * x.indices.shape[0]))) + 63) // 64 * 64)
            oidx, n_out, ohash, _ = ops.sparse_downsample_sites(x.indices, n_in, self.kernel_size, self.stride, self.padding, oshape,
                                                                 cap, err_flag=x.err)
            nbr, tm = ops.sparse_rulebook(oidx, n_out, self.kernel_size, self.stride, self.padding, in_hash)
            out = SparseConvTensor(None, oidx, oshape, x.batch_size, n_dev=n_out, err=x.err, growth=x.growth[1:])
            out.indice_dict["__hash__"] = (ops.SiteHash(ohash.capacity, oshape, oidx.device, ohash.keys, ohash.vals), n_out)
        else:
            oshape = [(d + 2 * p - k) // s + 1 for d, k, s, p in zip(x.spatial_shape, self.kernel_size, self.stride, self.padding)]
            cells = x.batch_size * int(np.prod(oshape))
            kv = int(np.prod(self.kernel_size))
            cap = min(cells, x.indices.shape[0] * min(kv, 8))
            cap = max(64, (cap + 63) // 64 * 64)
            oidx, n_out, ohash, err = ops.sparse_downsample_sites(x.indices, n_in, self.kernel_size, self.stride, self.padding, oshape, cap)
            nbr, tm = ops.sparse_rulebook(oidx, n_out, self.kernel_size, self.stride, self.padding, in_hash)
            m = int(n_out.item())  # API contract: exact row count (host sync; the fused engine avoids it)
            if int(err.item()):
                raise RuntimeError("sparse conv output capacity overflow")
            oidx, nbr, tm = oidx[:m].contiguous(), nbr[:, :m].contiguous(), tm[:(m + 15) // 16].contiguous()
            out = SparseConvTensor(None, oidx, oshape, x.batch_size)
            n_out = torch.tensor([m], dtype=torch.int32, device=oidx.device)
            out.indice_dict["__hash__"] = (ops.SiteHash(ohash.capacity, oshape, oidx.device, ohash.keys, ohash.vals), n_out)
        feats = x.features.float().contiguous()
        if torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad):
            subm_sym = self.subm and all(k % 2 == 1 for k in self.kernel_size)
            out.features = IndiceConvFunction.apply(feats, self.weight, self.bias, nbr, tm, n_out, n_in if x.n_dev is not None else None,
                                                    subm_sym, (plan, plan.generation) if plan is not None else None)
        else:
            out.features = ops.sparse_conv(feats, nbr, tm, n_out, self._wpk(), self.in_channels, self.out_channels, None,
                                           self.bias, relu=False)
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseSequential(SparseModule):
    """nn.Sequential over SparseConvTensor: non-sparse modules are applied to `.features` (spconv v1 behaviour)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for idx, module in enumerate(args):
            self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    # Train mode: BatchNorm1d + the ReLU that follows it run as the fused HIP passes of csrc/bn_train.hip (statistics, normalise +
    # ReLU; backward likewise) instead of the torch modules (tests/test_bn_train_gpu.py). False restores the torch modules.
    FUSED_BN_TRAIN = True

    def forward(self, input):
        mods = list(self._modules.values())
        skip = False
        for pos, module in enumerate(mods):
            if skip:
                skip = False
                continue
            if isinstance(module, SparseModule):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if (self.FUSED_BN_TRAIN and module.training and isinstance(module, torch.nn.BatchNorm1d)
                            and module.track_running_stats and module.affine and input.features.is_cuda):
                        relu = pos + 1 < len(mods) and isinstance(mods[pos + 1], torch.nn.ReLU)
                        n_dev = input.n_dev if input.n_dev is not None else torch.tensor(
                            [input.features.shape[0]], dtype=torch.int32, device=input.features.device)
                        input.features = ops.bn_relu_train(input.features, n_dev, module, relu=relu)
                        skip = relu
                    elif (input.n_dev is not None and getattr(module, "training", False)
                          and isinstance(module, torch.nn.modules.batchnorm._BatchNorm)):
                        # a torch BatchNorm in train mode would take its batch statistics over the padding rows as well
                        raise RuntimeError("capacity mode needs the fused train-mode BatchNorm (SparseSequential.FUSED_BN_TRAIN, "
                                           "affine BatchNorm1d with running statistics on the device)")
                    else:
                        input.features = module(input.features)
            else:
                input = module(input)
        return input


from . import utils  # noqa: E402,F401  (`from spconv.utils import rbbox_iou, rbbox_intersection`, box_np_ops.py:9)
