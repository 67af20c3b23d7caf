"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the sparse 3-D convolution stack (SpMiddleFHD).

PARITY UNPINNED: the arithmetic of this stage lives in the third-party package `spconv`
(traveller59/spconv, v1.x API: spconv.SubMConv3d / SparseConv3d / SparseConvTensor, version not
pinned by the reference: requirements.txt:28 says just `spconv`), which is neither vendored in
/root/reference nor installable here. This file restates spconv v1's published semantics and is
anchored on the reference's call sites (det3d/models/backbones/scn.py:92-189):
  * SubMConv3d(k=3): output sites == input sites; out[p] = sum_k W[k]^T in[p + k - 1]  (centred
    window, `padding` ignored, cross-correlation, no bias here)
  * SparseConv3d(k, s, p): out spatial = (D + 2p - k)//s + 1 per dim; an output site exists iff
    at least one active input lies in its receptive field; out[o] = sum_k W[k]^T in[o*s - p + k]
  * weights are [kz,ky,kx,Cin,Cout]; SparseSequential applies BatchNorm1d/ReLU to .features
  * .dense() scatters to (B,C,D,H,W); scn.py:186-187 then views it as (B, C*D, H, W)
The algorithm mirrors spconv's: per kernel offset gather -> mm -> scatter-add, float32.
It is cross-checked against torch.nn.functional.conv3d on the densified grid in the tests.
"""
import numpy as np
import torch


def _triple(v):
    return [int(v)] * 3 if isinstance(v, (int, np.integer)) else [int(x) for x in v]


def out_spatial(shape, ksize, stride, padding):
    return [(d + 2 * p - k) // s + 1 for d, k, s, p in zip(shape, _triple(ksize), _triple(stride), _triple(padding))]


def _lin(idx, shape):
    idx = idx.astype(np.int64)
    return ((idx[:, 0] * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


def rulebook(indices, spatial_shape, ksize, stride, padding, subm):
    """indices (N,4) int32 [b,z,y,x]. Returns out_indices (M,4) int32 (ascending linear order for a
    regular conv, the input order for subm), out_shape, and per-offset (in_rows, out_rows)."""
    ks, st, pd = _triple(ksize), _triple(stride), _triple(padding)
    idx = np.asarray(indices, np.int64)
    if subm:
        oshape = list(spatial_shape)
        pd = [k // 2 for k in ks]
        st = [1, 1, 1]
    else:
        oshape = out_spatial(spatial_shape, ks, st, pd)
    offs = [(a, b, c) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])]
    cand = []
    for (a, b, c) in offs:
        o = idx[:, 1:] + np.array(pd) - np.array([a, b, c])
        ok = np.ones(len(idx), bool)
        for d in range(3):
            ok &= (o[:, d] % st[d] == 0)
        o = o // np.array(st)
        for d in range(3):
            ok &= (o[:, d] >= 0) & (o[:, d] < oshape[d])
        cand.append((ok, np.concatenate([idx[:, :1], o], 1)))
    if subm:
        out_idx = idx
        lin_out = _lin(out_idx, oshape)
        order = np.argsort(lin_out, kind="stable")
        sorted_lin = lin_out[order]
    else:
        allo = np.concatenate([c[1][c[0]] for c in cand], 0)
        lin = np.unique(_lin(allo, oshape))
        sorted_lin = lin
        order = np.arange(len(lin))
        x = lin % oshape[2]
        y = (lin // oshape[2]) % oshape[1]
        z = (lin // (oshape[2] * oshape[1])) % oshape[0]
        b = lin // (oshape[2] * oshape[1] * oshape[0])
        out_idx = np.stack([b, z, y, x], 1)
    pairs = []
    for ok, o in cand:
        rows_in = np.nonzero(ok)[0]
        lo = _lin(o[rows_in], oshape)
        pos = np.searchsorted(sorted_lin, lo)
        pos = np.clip(pos, 0, len(sorted_lin) - 1)
        hit = sorted_lin[pos] == lo if len(sorted_lin) else np.zeros(len(lo), bool)
        pairs.append((rows_in[hit], order[pos[hit]]))
    return out_idx.astype(np.int32), oshape, pairs


def sparse_conv(features, indices, spatial_shape, weight, ksize, stride=1, padding=0, subm=False, rb=None):
    """features (N,Cin) f32 torch; weight (kz,ky,kx,Cin,Cout) torch. Returns (out_features, out_indices, out_shape, rb)."""
    if rb is None:
        rb = rulebook(indices, spatial_shape, ksize, stride, padding, subm)
    out_idx, oshape, pairs = rb
    w = weight.reshape(-1, weight.shape[-2], weight.shape[-1])
    out = torch.zeros((out_idx.shape[0], w.shape[-1]), dtype=torch.float32)
    for k, (ri, ro) in enumerate(pairs):
        if len(ri) == 0:
            continue
        out.index_add_(0, torch.from_numpy(ro.astype(np.int64)), features[torch.from_numpy(ri.astype(np.int64))] @ w[k])
    return out, out_idx, oshape, rb


def bn_relu(x, bn, relu=True, eps=1e-3, training=False):
    """bn = dict(weight, bias, running_mean, running_var): BatchNorm1d (scn.py:103-104 eps=1e-3); eval mode by default,
    training=True normalises with the batch statistics (the SE-SSD training step runs both nets in train mode,
    trainer_sessd.py:321-322) without touching the running statistics."""
    if training:
        y = torch.nn.functional.batch_norm(x, None, None, bn["weight"], bn["bias"], True, 0.0, eps)
    else:
        y = torch.nn.functional.batch_norm(x, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, eps)
    return torch.relu(y) if relu else y


def dense(features, indices, spatial_shape, batch_size):
    C = features.shape[1]
    out = torch.zeros([batch_size] + list(spatial_shape) + [C], dtype=features.dtype)
    i = torch.from_numpy(np.asarray(indices, np.int64))
    out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = features
    return out.permute(0, 4, 1, 2, 3).contiguous()


# scn.py:106-148: (kind, cin, cout, ksize, stride, padding, indice_key)
SPMIDDLE_FHD_LAYERS = [
    ("subm", 4, 16, 3, 1, 0, "subm0"), ("subm", 16, 16, 3, 1, 0, "subm0"),
    ("conv", 16, 32, 3, 2, 1, None),
    ("subm", 32, 32, 3, 1, 0, "subm1"), ("subm", 32, 32, 3, 1, 0, "subm1"),
    ("conv", 32, 64, 3, 2, 1, None),
    ("subm", 64, 64, 3, 1, 0, "subm2"), ("subm", 64, 64, 3, 1, 0, "subm2"), ("subm", 64, 64, 3, 1, 0, "subm2"),
    ("conv", 64, 64, 3, 2, [0, 1, 1], None),
    ("subm", 64, 64, 3, 1, 0, "subm3"), ("subm", 64, 64, 3, 1, 0, "subm3"), ("subm", 64, 64, 3, 1, 0, "subm3"),
    ("conv", 64, 64, (3, 1, 1), (2, 1, 1), 0, None),
]


def spmiddle_fhd(voxel_features, coors, batch_size, input_shape, weights, bns, layers=None, return_levels=False,
                 training=False):
    """SpMiddleFHD.forward (scn.py:176-189). weights[i] (kz,ky,kx,Cin,Cout), bns[i] dict. input_shape = [x,y,z] grid."""
    layers = layers or SPMIDDLE_FHD_LAYERS
    shape = [int(v) for v in (np.array(input_shape[::-1]) + [1, 0, 0])]
    feat = voxel_features
    idx = np.asarray(coors, np.int32)
    cache = {}
    levels = []
    for i, (kind, cin, cout, ks, st, pd, key) in enumerate(layers):
        if kind == "subm":
            rb = cache.get(key)
            feat, idx2, shape2, rb = sparse_conv(feat, idx, shape, weights[i], ks, 1, 0, True, rb)
            cache[key] = rb
        else:
            feat, idx2, shape2, rb = sparse_conv(feat, idx, shape, weights[i], ks, st, pd, False)
        idx, shape = idx2, shape2
        feat = bn_relu(feat, bns[i], training=training)
        levels.append((feat, idx, list(shape)))
    d = dense(feat, idx, shape, batch_size)
    N, C, D, H, W = d.shape
    out = d.view(N, C * D, H, W)
    return (out, levels) if return_levels else out
