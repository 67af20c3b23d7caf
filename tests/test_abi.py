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
    # Winograd stream-K over weight sets: the batch must split evenly
    assert lib.sessd_conv3x3_winograd_sk_sets(None, 3, 2, 128, 8, 8, None, None, 128, None, None, 1, None, None, 0, 0, 8, None) == -1
    assert lib.sessd_conv3x3_winograd_sk_sets(None, 2, 2, 128, 7, 8, None, None, 128, None, None, 1, None, None, 0, 0, 8, None) == -1
    # two transposed convs in one launch / fused SSFA tail + heads
    assert lib.sessd_deconv2d_s2_mfma_pair(None, 1, 12, 8, 8, None, None, None, None, None, None, None, 32, None, None, None, None, 1, None, None,
                                           4, None) == -1
    assert lib.sessd_ssfa_fuse_head(None, None, None, None, 1.0, 0.0, 1.0, 0.0, 1, 100, 64, None, None, None, 22, None, None) == -1
    assert lib.sessd_ssfa_fuse_head(None, None, None, None, 1.0, 0.0, 1.0, 0.0, 1, 128, 64, None, None, None, 21, None, None) == -1
