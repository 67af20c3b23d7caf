"""HIP predict / rotated NMS vs the CPU oracle (oracle/postprocess.py + oracle/rotate_nms.c).

Selection (which anchors survive, in which order) must be IDENTICAL unless the oracle reports a pair
within 1e-4 of the NMS threshold or a score within 1e-6 of the score threshold; kept boxes must agree
to 1e-4 (m / rad) and scores to 1e-6 relative."""
import numpy as np
import pytest
import torch

from oracle import capi, postprocess as pp
from sessd_hip import ops, synth

pytestmark = pytest.mark.gpu

H, W = 200, 176


def _make_head(seed, n_hot, H=H, W=W, cls_bias=-4.0):
    rng = np.random.RandomState(seed)
    P = H * W
    head = np.zeros((22, P), np.float32)
    head[:14] = rng.normal(0, 0.25, (14, P))
    head[14:16] = cls_bias + rng.normal(0, 0.5, (2, P))
    head[16:20] = rng.normal(0, 1.0, (4, P))
    head[20:22] = rng.uniform(-0.2, 1.0, (2, P))
    # hot blobs: clusters of neighbouring locations with high class logits (real NMS work)
    for _ in range(n_hot):
        cy, cx = rng.randint(5, H - 5), rng.randint(5, W - 5)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                p = (cy + dy) * W + cx + dx
                head[14 + rng.randint(2), p] = rng.uniform(-0.5, 3.0)
    return head


def _split(head):
    P = head.shape[1]
    box = head[:14].reshape(2, 7, P).transpose(2, 0, 1).reshape(-1, 7)
    cls = head[14:16].T.reshape(-1)
    dirl = head[16:20].reshape(2, 2, P).transpose(2, 0, 1).reshape(-1, 2)
    iou = head[20:22].T.reshape(-1)
    return box, cls, dirl, iou


@pytest.mark.parametrize("seed,n_hot,use_frustum", [(0, 40, True), (1, 150, False), (2, 0, True), (3, 400, True)])
def test_predict_vs_oracle(dev, seed, n_hot, use_frustum):
    head = _make_head(seed, n_hot)
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    cal = synth.kitti_calib()
    fr = pp.get_valid_frustum(cal["rect"], cal["Trv2c"], cal["P2"], cal["image_shape"]) if use_frustum else None
    box, cls, dirl, iou = _split(head)
    want, dbg = pp.predict_frame(box, cls, dirl, iou, anchors, fr, return_debug=True)
    out = ops.predict(torch.from_numpy(head[None]).to(dev), torch.from_numpy(anchors).to(dev),
                      None if fr is None else torch.from_numpy(fr[None].copy()).to(dev))
    n = int(out["count"][0].item())
    gb = out["box"][0, :n].cpu().numpy()
    gs = out["score"][0, :n].cpu().numpy()
    sc = pp.sigmoid32(cls)
    borderline = int((np.abs(sc - 0.3) < 1e-6).sum())
    if n != len(want["scores"]) or not np.allclose(gs, want["scores"], rtol=1e-5, atol=1e-7):
        assert dbg.get("near_threshold_pairs", 0) > 0 or borderline > 0, (n, len(want["scores"]), dbg)
        pytest.skip("selection differs only through a pair/score sitting on a threshold")
    assert np.abs(gb - want["box3d_lidar"]).max() < 1e-4 if n else True
    assert (out["label"][0, :n].cpu().numpy() == 0).all()
    print("candidates", dbg["num_candidates"], "kept", n)


def test_predict_batch_and_many_candidates(dev):
    """B=3 with one frame above the 2048-key sort window (running top-k) and one empty frame."""
    heads = [_make_head(5, 60), _make_head(6, 0, cls_bias=-0.2), _make_head(7, 0, cls_bias=-30.0)]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    out = ops.predict(torch.from_numpy(np.stack(heads)).to(dev), torch.from_numpy(anchors).to(dev), None)
    counts = out["count"].cpu().numpy()
    for b, head in enumerate(heads):
        box, cls, dirl, iou = _split(head)
        want, dbg = pp.predict_frame(box, cls, dirl, iou, anchors, None, return_debug=True)
        n = int(counts[b])
        if b == 1:
            assert dbg["num_candidates"] > 2048
        if n != len(want["scores"]):
            assert dbg.get("near_threshold_pairs", 0) > 0
            continue
        assert np.allclose(out["score"][b, :n].cpu().numpy(), want["scores"], rtol=1e-5, atol=1e-7)
        if n:
            assert np.abs(out["box"][b, :n].cpu().numpy() - want["box3d_lidar"]).max() < 1e-4
    assert counts[2] == 0


@pytest.mark.parametrize("n,thresh", [(1, 0.01), (100, 0.01), (700, 0.1), (1000, 0.5), (2500, 0.3)])
def test_rotate_nms_sorted_vs_oracle(dev, n, thresh):
    b = synth.clustered_boxes7(n, seed=n)
    rng = np.random.RandomState(n)
    score = np.sort(rng.uniform(0.3, 1, n).astype(np.float32))[::-1]
    dets = np.concatenate([b[:, [0, 1, 3, 4, 6]], score[:, None]], 1).astype(np.float32)
    want, near = capi.rotate_nms_cc(dets, thresh, order=np.arange(n, dtype=np.int32))
    keep, num = ops.rotate_nms_sorted(torch.from_numpy(np.ascontiguousarray(dets[:, :5])).to(dev), thresh, 100)
    k = int(num.item())
    got = keep[:k].cpu().numpy()
    if not np.array_equal(got, want[:100]):
        assert near > 0, "rotated NMS differs without any near-threshold pair"
