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


def test_fused_forms_equal_the_plain_call(dev):
    """sessd_predict_fused's two optional fusions give what the separate launches give, bit for bit:
    * the score-filter keys produced INSIDE the head launch (sessd_ssfa_fuse_head_keys) -- same detections as the call that
      filters the stored head tensor itself, and the same KEY SET as score_filter_kernel would append;
    * the frame's detection record written by the call's last launch == sessd_pack_detections of the outputs (ring rule:
      slot = (cursor + b) % capacity, cursor += batch), also across a wrap of the ring."""
    torch.manual_seed(0)
    B, C = 2, 128
    x0 = torch.randn(B, C, H, W, device=dev)
    x1 = torch.randn(B, C, H, W, device=dev)
    w0, w1 = torch.randn(C, device=dev) * 0.05, torch.randn(C, device=dev) * 0.05
    hw = torch.randn(22, C, device=dev) * 0.08
    hb = torch.randn(22, device=dev) * 0.1
    hb[14:16] = -2.2  # a few hundred to a few thousand candidates above the score threshold
    anchors = torch.from_numpy(pp.create_anchors_3d_range().reshape(-1, 7)).to(dev)
    keys = torch.zeros((B, 2 * H * W), dtype=torch.int64, device=dev)
    kcnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    head = ops.ssfa_fuse_head(x0, x1, w0, w1, 1.1, 0.05, 0.9, -0.03, hw, hb, score_thresh=0.3, keys=keys, key_count=kcnt)
    head_plain = ops.ssfa_fuse_head(x0, x1, w0, w1, 1.1, 0.05, 0.9, -0.03, hw, hb)
    assert torch.equal(head, head_plain)
    # the key set of the stand-alone filter, recomputed on the host from the stored tensor in float32
    hc = head.cpu().numpy()
    for b in range(B):
        s = pp.sigmoid32(hc[b, 14:16].T.reshape(-1))
        n_want = int((s >= np.float32(0.3)).sum())
        n_got = int(kcnt[b].item())
        assert abs(n_got - n_want) <= int((np.abs(s - 0.3) < 1e-6).sum()) and n_got > 100
        aid = (keys[b, :n_got].cpu().numpy() & 0xFFFFFFFF).astype(np.int64)
        assert len(np.unique(aid)) == n_got and (s[aid] >= 0.3 - 1e-6).all()
    plain = ops.predict(head, anchors, None)
    cap = 3
    rec = torch.full((cap, 100, 9), -7.0, device=dev)
    rcnt = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    cur = torch.zeros((1,), dtype=torch.int32, device=dev)
    for rep in range(2):  # second call wraps: slots 2, 0
        fused = ops.predict(head, anchors, None, keys=keys, key_count=kcnt, records=(rec, rcnt, cur))
        for k in ("box", "score", "label", "count"):
            assert torch.equal(fused[k][:, :int(plain["count"].max())] if k != "count" else fused[k],
                               plain[k][:, :int(plain["count"].max())] if k != "count" else plain[k]), k
        assert int(cur.item()) == 2 * (rep + 1)
        for b in range(B):
            slot = (2 * rep + b) % cap
            n = int(plain["count"][b].item())
            assert int(rcnt[slot].item()) == n and n > 5
            r = rec[slot].cpu()
            assert torch.equal(r[:n, :7], plain["box"][b, :n].cpu()) and torch.equal(r[:n, 7], plain["score"][b, :n].cpu())
            assert float(r[:n, 8].abs().max()) == 0 and float(r[n:].abs().max()) == 0


def test_fill_multi(dev):
    a = torch.full((1000003,), 5, dtype=torch.int32, device=dev)[:1000000]   # not a multiple of 4096 words
    b = torch.full((4096 * 3,), 5, dtype=torch.int32, device=dev)
    c = torch.full((7,), 5.0, dtype=torch.float32, device=dev)
    guard = a.clone()
    ops.fill_multi([(a[:999996], 0x7F7F7F7F), (b, 0), (c[:4], 0x3F800000)])
    assert int((a[:999996] != 0x7F7F7F7F).sum()) == 0 and torch.equal(a[999996:], guard[999996:])
    assert int(b.abs().sum()) == 0
    assert torch.equal(c.cpu(), torch.tensor([1.0, 1, 1, 1, 5, 5, 5]))
