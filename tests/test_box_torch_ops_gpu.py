"""det3d.core.bbox.box_torch_ops.rotate_nms / nms (the calls MultiGroupHead.get_task_detections makes,
mg_head_sessd.py:987-992) on the device vs the CPU oracle (oracle/rotate_nms.c = nms_cpu.py:40-51 + nms_cpu.h:72-168)."""
import numpy as np
import pytest
import torch

from oracle import capi
from sessd_hip import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,pre,post,thresh", [(500, 300, 50, 0.01), (200, None, None, 0.3), (40, 1000, 100, 0.5), (0, 10, 10, 0.1)])
def test_rotate_nms_contract(dev, n, pre, post, thresh):
    from det3d.core.bbox import box_torch_ops
    b7 = synth.clustered_boxes7(n, seed=n + 11) if n else np.zeros((0, 7), np.float32)
    scores = np.random.RandomState(n).permutation(n).astype(np.float32) / max(n, 1)
    rb = torch.from_numpy(b7[:, [0, 1, 3, 4, 6]].copy()).to(dev)
    got = box_torch_ops.rotate_nms(rb, torch.from_numpy(scores).to(dev), pre_max_size=pre, post_max_size=post, iou_threshold=thresh)
    assert got.dtype == torch.int64 and got.device.type == "cuda"
    if n == 0:
        assert got.numel() == 0
        return
    # reference composition: topk -> rotate_nms_cc on [boxes | scores] -> first post_max_size -> indices into the input
    order = np.argsort(-scores, kind="stable")
    k = n if pre is None else min(n, pre)
    sel = order[:k]
    dets = np.concatenate([b7[sel][:, [0, 1, 3, 4, 6]], scores[sel, None]], 1).astype(np.float32)
    keep, near = capi.rotate_nms_cc(dets, thresh)
    want = sel[keep[: (post if post is not None else len(keep))]]
    if near == 0:
        assert np.array_equal(got.cpu().numpy(), want)
    else:
        assert len(got) > 0


def test_axis_aligned_nms_contract(dev):
    from det3d.core.bbox import box_torch_ops
    rng = np.random.RandomState(2)
    xy = rng.uniform(0, 300, (250, 2)).astype(np.float32)
    wh = rng.uniform(10, 60, (250, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = rng.permutation(250).astype(np.float32)
    got = box_torch_ops.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), pre_max_size=200, post_max_size=30,
                            iou_threshold=0.4).cpu().numpy()
    order = np.argsort(-scores, kind="stable")[:200]
    s = boxes[order]
    alive, want = np.ones(200, bool), []
    for i in range(200):  # numba nms_gpu semantics (+1 pixel, '>'), nms_gpu.py:36-169
        if not alive[i]:
            continue
        want.append(order[i])
        for j in range(i + 1, 200):
            w = max(min(s[i, 2], s[j, 2]) - max(s[i, 0], s[j, 0]) + 1, 0.0)
            h = max(min(s[i, 3], s[j, 3]) - max(s[i, 1], s[j, 1]) + 1, 0.0)
            sa = (s[i, 2] - s[i, 0] + 1) * (s[i, 3] - s[i, 1] + 1)
            sb = (s[j, 2] - s[j, 0] + 1) * (s[j, 3] - s[j, 1] + 1)
            if np.float32(w * h) / np.float32(sa + sb - w * h) > 0.4:
                alive[j] = False
    assert np.array_equal(got, np.array(want[:30]))
