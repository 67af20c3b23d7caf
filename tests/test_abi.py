"""The C-ABI library loads and exports every symbol include/sessd_hip.h declares (no compute)."""
import os
import re

from conftest import ROOT


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "sessd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sessd_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    import sessd_hip
    names = _header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(sessd_hip.lib, n), "libsessd_hip.so does not export %s" % n
        assert n in sessd_hip.SIGNATURES, "sessd_hip/_lib.py has no binding for %s" % n
    for n in sessd_hip.SIGNATURES:
        assert n in names, "%s is bound but not declared in include/sessd_hip.h" % n
    assert b"gfx950" in sessd_hip.lib.sessd_version()


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "se-ssd_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from sessd_hip import ops
    with pytest.raises(ValueError):
        ops.boxes_pairwise(1, torch.zeros(2, 5), torch.zeros(2, 5))


def test_dense_entry_points_reject_bad_shapes_before_touching_the_device():
    """Argument checks of the dense-conv entry points added in round 2 come before any device call: they can be exercised here
    (null pointers, no GPU). SESSD_EINVAL = -1; workspace-size queries return 0 for bad arguments."""
    import sessd_hip
    lib = sessd_hip.lib
    # LDS-tiled stream-K conv: cin % 16, class count, workgroups % 8, 32-bit buffer offsets
    sk = lambda cin, nclass, wgs, hw=16: lib.sessd_conv2d_sk(None, 1, cin, hw, hw, nclass, None, None, None, None, 1, hw, hw, None, 32, hw, hw, 1,
                                                             None, None, None, None, 1, None, None, 0, wgs, None)
    assert sk(24, 1, 8) == -1 and sk(32, 5, 8) == -1 and sk(32, 1, 12) == -1 and sk(4096, 1, 8, hw=1024) == -1
    assert lib.sessd_conv2d_sk_workspace_bytes(0, 8, 8, 32, 1, 8) == 0 and lib.sessd_conv2d_sk_workspace_bytes(1, 8, 8, 32, 5, 8) == 0
    # explicit workgroup count: pure arithmetic (counters of nclass * batch * pixel tiles * cout groups + two 64 KB slots per workgroup)
    assert lib.sessd_conv2d_sk_workspace_bytes(2, 100, 88, 256, 1, 256) == 256 * ((2 * 69 * 2 * 4 + 255) // 256) + 2 * 256 * 65536
    assert lib.sessd_conv2d_sk_pack(None, 1, 1, None, 1, 32, 24, None, None) == -1
    # its active-tile mode: a tile list with its device count, even tile space, the WHOLE batch inside 32-bit offsets
    import ctypes
    one = ctypes.c_int32(1)
    ska = lambda tl, nl, cap, hw=16, batch=1, cin=32: lib.sessd_conv2d_sk_active(None, batch, cin, hw, hw, 1, None, None, None, None, 1, hw, hw, None, 32,
                                                                                 hw, hw, 1, None, None, None, None, 1, None, tl, nl, cap, 1, None, 0, 8, None)
    ptr = ctypes.addressof(one)
    assert ska(None, ptr, 4) == -1 and ska(ptr, None, 4) == -1 and ska(ptr, ptr, 0) == -1 and ska(ptr, ptr, 4, hw=15) == -1
    assert ska(ptr, ptr, 4, hw=1024, batch=32, cin=16) == -1
    # tile activity: a stride-2 step that takes a slot needs an output of even size; at most eight slots; step 4 only behind a step 3
    steps = lambda *v: (ctypes.c_int32 * len(v))(*v)
    act = lambda st, h=200, w=176: lib.sessd_bev_tile_activity(ptr, ptr, 1, 1, h, w, st, len(st), ptr, ptr, ptr, 1, None, 0, None)
    assert act(steps(2), h=202) == -1 and act(steps(0, 0, 0, 2, 0, 0, 0, 0, 0)) == -1 and act(steps(4)) == -1 and act(steps(0, 4)) == -1 and act(steps(3), w=192) == -1
    assert act(steps(3, 4, 0), h=200, w=100) == -1   # the doubled map must fit the kernel's pixel rows
    assert lib.sessd_fill_inactive_tiles(None, 13, 1, None) == -1
    # Winograd stream-K over weight sets: the batch must split evenly
    assert lib.sessd_conv3x3_winograd_sk_sets(None, 3, 2, 128, 8, 8, None, None, 128, None, None, 1, None, None, 0, 0, 8, None) == -1
    assert lib.sessd_conv3x3_winograd_sk_sets(None, 2, 2, 128, 7, 8, None, None, 128, None, None, 1, None, None, 0, 0, 8, None) == -1
    # two transposed convs in one launch / fused SSFA tail + heads
    assert lib.sessd_deconv2d_s2_mfma_pair(None, 1, 12, 8, 8, None, None, None, None, None, None, None, 32, None, None, None, None, 1, None, None,
                                           4, None) == -1
    assert lib.sessd_ssfa_fuse_head(None, None, None, None, 1.0, 0.0, 1.0, 0.0, 1, 100, 64, None, None, None, 22, None, None) == -1
    assert lib.sessd_ssfa_fuse_head(None, None, None, None, 1.0, 0.0, 1.0, 0.0, 1, 128, 64, None, None, None, 21, None, None) == -1


def test_sparse_entry_points_reject_what_their_kernels_cannot_cover():
    """Round-2 advisor / review items, checked before any device call (null device pointers, no GPU):
    * sessd_sparse_chain_rulebooks: chain_rulebook_kernel covers at most 12 (kz, ky) pairs and 3 taps in x -- a (5,5,1) or
      (1,1,5) kernel passed the old volume <= 32 check and would have left neighbour-table rows unwritten;
    * sessd_sparse_hash_build: one batch element's D*H*W must stay below the empty marker 0x7F7F7F7F of the 31-bit keys."""
    import ctypes as C
    import sessd_hip
    from sessd_hip._lib import ChainLevel, RulebookJob
    lib = sessd_hip.lib
    lv = (ChainLevel * 1)()
    lv[0].ksize[:] = [3, 3, 3]
    lv[0].stride[:] = [2, 2, 2]
    lv[0].pad[:] = [1, 1, 1]
    lv[0].out_dims[:] = [21, 800, 704]
    lv[0].cap = 1024
    dims0 = (C.c_int * 3)(41, 1600, 1408)
    ws = (C.c_char * 16)()

    def rulebooks(ks):
        job = (RulebookJob * 1)()
        job[0].in_level, job[0].out_level = 1, 1
        job[0].ksize[:] = ks
        job[0].stride[:] = [1, 1, 1]
        job[0].pad[:] = [k // 2 for k in ks]
        job[0].nbr, job[0].tile_mask = 16, 16  # non-null, never dereferenced: the call must fail first
        return lib.sessd_sparse_chain_rulebooks(None, None, 1024, None, None, 1024, C.cast(dims0, C.c_void_p), 1, 1, C.cast(lv, C.c_void_p),
                                                C.cast(ws, C.c_void_p), 1, C.cast(job, C.c_void_p), None)

    assert rulebooks([5, 5, 1]) == -1 and rulebooks([1, 1, 5]) == -1 and rulebooks([4, 4, 2]) == -1 and rulebooks([0, 3, 3]) == -1
    big = (C.c_int * 3)(2048, 2048, 1024)
    assert lib.sessd_sparse_hash_build(None, None, 16, C.cast(big, C.c_void_p), None, None, 64, None) == -1
    assert lib.sessd_sparse_hash_build(None, None, 16, None, None, None, 64, None) == -1
    assert lib.sessd_sparse_hash_build(None, None, 16, C.cast(dims0, C.c_void_p), None, None, 48, None) == -1  # capacity not a power of two


def test_round3_training_entry_points_reject_bad_arguments_before_touching_the_device():
    """Argument checks of the training-path entry points of round 3, before any device call (null pointers, no GPU):
    the Winograd-domain weight gradient's shape rule and workspace arithmetic, the fused SSFA attention tail, the train-mode
    BatchNorm layouts, the batched packers, the channel sum and the host box functions."""
    import ctypes as C
    import numpy as np
    import sessd_hip
    lib = sessd_hip.lib
    wb = lib.sessd_conv3x3_wgrad_winograd_workspace_bytes
    # covered: cin, cout % 64, even H, W >= 16;  64 chunks x 16 xi x cin x cout floats for the SSFA layers at batch 4
    assert wb(4, 128, 128, 200, 176) == 64 * 16 * 128 * 128 * 4 and wb(4, 256, 256, 100, 88) == 16 * 16 * 256 * 256 * 4
    assert wb(1, 48, 128, 64, 64) == 0 and wb(1, 128, 128, 63, 64) == 0 and wb(1, 128, 128, 64, 12) == 0 and wb(0, 128, 128, 64, 64) == 0
    assert lib.sessd_conv3x3_wgrad_winograd(None, 1, 128, 64, 64, None, 128, None, None, 0, None) == -1        # null tensors
    assert lib.sessd_conv3x3_wgrad_winograd(16, 1, 128, 64, 64, 16, 128, 16, None, 0, None) == -2               # workspace too small
    # fused SSFA tail: channels % 4, plane % 4, all-or-none running statistics
    sb = lib.sessd_ssfa_fuse_train_workspace_bytes
    assert sb(4, 128, 35200) > 4352 and sb(4, 126, 35200) == 0 and sb(4, 128, 35202) == 0 and sb(0, 128, 64) == 0
    assert lib.sessd_ssfa_fuse_train_fwd(None, None, 1, 128, 64, None, None, None, None, None, None, 1e-3, 0.01, None, None, None, None, None,
                                         None, None, None, 0, None) == -1
    assert lib.sessd_ssfa_fuse_train_fwd(16, 16, 1, 128, 64, 16, 16, None, None, None, None, 1e-3, 0.01, 16, None, None, None, 16, 16, 16, 16,
                                         1 << 20, None) == -1   # one of the four running-statistics pointers
    # train-mode BatchNorm: power-of-two channels <= 256 (sparse layout), plane % 4 and <= 1024 channels (dense layout)
    assert lib.sessd_bn_relu_train_fwd(None, None, 64, 24, None, None, 1e-3, 0.01, 1, None, None, None, None, None, None, 0, None) == -1
    assert lib.sessd_bn_relu_train_fwd(None, None, 64, 16, None, None, 1e-3, 0.01, 1, None, None, None, None, None, None, 0, None) == -2
    assert lib.sessd_bn2d_relu_train_fwd(None, 1, 8, 30, None, None, 1e-3, 0.01, 1, None, None, None, None, None, None, 0, None) == -1
    assert lib.sessd_bn2d_relu_train_bwd_x(None, None, 1, 2048, 64, None, None, None, None, 1, None, None, None, None, 0, None) == -1
    assert lib.sessd_bn2d_relu_train_bwd(None, None, None, 1, 8, 64, None, None, None, 1, None, 16, 16, 16, 1 << 20, None) == -1   # relu without y
    assert lib.sessd_nchw_channel_sum(None, 1, 8, 30, None, None, 0, None) == -1
    assert lib.sessd_bn2d_relu_train_workspace_bytes(128) == 4096 + 128 * 16 * 2 * 8
    assert lib.sessd_bn_relu_train_workspace_bytes(64) == 256 + 128 * 2 * 64 * 8
    # batched packers, adjoint pack
    assert lib.sessd_dense_pack_batch(None, 3, 10, None) == -1 and lib.sessd_sparse_pack_batch(16, 0, 10, None) == -1
    assert lib.sessd_sparse_pack_weight_adjoint(None, 27, 24, 64, 1, None, None) == -1
    # host box functions run here: two overlapping unit squares, one far away
    sq = np.array([[[0, 0], [0, 1], [1, 1], [1, 0]]], dtype=np.float64)
    qs = np.concatenate([sq + 0.5, sq + 5.0], 0)
    out = np.zeros((1, 2), dtype=np.uint8)
    assert lib.sessd_box_collision_host(sq.ctypes.data, 1, qs.ctypes.data, 2, 0, 1, out.ctypes.data) == 0 and out.tolist() == [[1, 0]]
    assert lib.sessd_box_collision_host(None, 1, qs.ctypes.data, 2, 0, 1, out.ctypes.data) == -1
    assert lib.sessd_noise_per_box_host(None, None, None, None, None, None, 2, 3, 0, None) == -1


def test_head_loss_entry_point_checks_its_arguments():
    """sessd_head_loss (round 4): configuration and pointer checks come before any device call (no GPU here)."""
    import ctypes as C
    import sessd_hip
    from sessd_hip._lib import HeadLossCfg, HeadLossNet
    lib = sessd_hip.lib
    c = HeadLossCfg()
    assert lib.sessd_head_loss_workspace_bytes(C.addressof(c)) == 0          # batch 0
    c.batch, c.num_anchors, c.pos_capacity, c.cons_capacity = 4, 70400, 4096, 2048
    c.pos_cls_weight = c.neg_cls_weight = 1.0
    c.smooth_l1_sigma = 3.0
    need = lib.sessd_head_loss_workspace_bytes(C.addressof(c))
    # counts per block + 12 doubles per block + the positive / candidate lists: a few MB, growing with the capacities
    assert 1 << 19 < need < 64 << 20
    c.cons_capacity = 4096
    assert lib.sessd_head_loss_workspace_bytes(C.addressof(c)) > need
    n = HeadLossNet()
    assert lib.sessd_head_loss(C.addressof(c), C.addressof(n), C.addressof(n), None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.sessd_head_loss(None, None, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    c.batch = 65
    assert lib.sessd_head_loss_workspace_bytes(C.addressof(c)) == 0
