"""The pybind-module surface `det3d.ops.nms.nms` and the numpy wrappers of `det3d.ops.nms.nms_cpu` on the
HIP kernels vs the CPU oracle: oracle/rotate_nms.c (nms_cpu.h:72-168 restated) and a plain-loop restatement of
nms_cpu.h:24-70 written here. Index lists must be identical."""
import numpy as np
import pytest

from oracle import capi
from sessd_hip import synth

pytestmark = pytest.mark.gpu


def _axis_nms_loop(boxes, order, thresh, eps, strict=False):
    """nms_cpu.h:24-70 as a plain loop (float32 arithmetic); strict=True is the '>' test of the numba / CUDA kernels."""
    b = boxes.astype(np.float32)
    e = np.float32(eps)
    area = (b[:, 2] - b[:, 0] + e) * (b[:, 3] - b[:, 1] + e)
    sup = np.zeros(len(b), bool)
    keep = []
    for _i in range(len(b)):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, len(b)):
            j = order[_j]
            if sup[j]:
                continue
            w = min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]) + e
            if w > 0:
                h = min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]) + e
                if h > 0:
                    inter = np.float32(w * h)
                    ovr = inter / (area[i] + area[j] - inter)
                    if (ovr > np.float32(thresh)) if strict else (ovr >= np.float32(thresh)):
                        sup[j] = True
    return keep


@pytest.mark.parametrize("n,thresh,eps", [(300, 0.5, 1.0), (300, 0.1, 0.0), (1, 0.5, 0.0), (0, 0.5, 0.0)])
def test_axis_aligned_family(dev, n, thresh, eps):
    from det3d.ops.nms import nms, nms_cpu
    rng = np.random.RandomState(n + 1)
    xy = rng.uniform(0, 60, (n, 2)).astype(np.float32)
    wh = rng.uniform(2, 12, (n, 2)).astype(np.float32)
    dets = np.concatenate([xy, xy + wh, rng.permutation(n).astype(np.float32)[:, None] / max(n, 1)], 1).astype(np.float32)
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    want = _axis_nms_loop(dets, order, thresh, eps)
    assert nms.non_max_suppression_cpu(dets, order, thresh, eps) == want
    if n:
        assert nms_cpu.nms_jit(dets, thresh, eps) == want
        if eps == 1.0:
            assert nms_cpu.nms_cc(dets, thresh) == want
        # the GPU variant (+1 convention, '>' instead of '>='): boxes pre-sorted, keep_out filled
        srt = np.ascontiguousarray(dets[order])
        keep_out = np.zeros(n, np.int64)
        cnt = nms.non_max_suppression(srt, keep_out, thresh, 0)
        want_gpu = _axis_nms_loop(srt, np.arange(n), thresh, 1.0, strict=True)
        assert cnt == len(want_gpu) and list(keep_out[:cnt]) == want_gpu


@pytest.mark.parametrize("n,thresh", [(400, 0.01), (400, 0.3), (64, 0.7), (0, 0.1)])
def test_rotate_nms_cc_and_corner_entry(dev, n, thresh):
    from det3d.core.bbox import box_np_ops
    from det3d.ops.nms import nms, nms_cpu
    b7 = synth.clustered_boxes7(n, seed=n + 3) if n else np.zeros((0, 7), np.float32)
    rng = np.random.RandomState(n)
    dets = np.concatenate([b7[:, [0, 1, 3, 4, 6]], rng.permutation(n).astype(np.float32)[:, None] / max(n, 1)], 1).astype(np.float32)
    got = nms_cpu.rotate_nms_cc(dets, thresh)
    want, near = capi.rotate_nms_cc(dets, thresh)
    if near == 0:
        assert list(got) == list(want)
    else:  # a pair within the oracle's margin of the threshold may flip: same prefix up to the first difference is enough
        assert len(got) > 0
    if n:
        order = dets[:, 5].argsort()[::-1].astype(np.int32)
        corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4]).astype(np.float32)
        got2 = nms.rotate_non_max_suppression_cpu(corners, order, None, thresh)
        assert list(got2) == list(got)
    with pytest.raises(TypeError):   # DI-NMS is built (tests/test_di_nms_gpu.py); the pybind signature has 14 arguments
        nms.IOU_weighted_rotate_non_max_suppression_cpu()
