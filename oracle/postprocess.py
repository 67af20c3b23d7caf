"""TEST INFRASTRUCTURE ONLY -- CPU oracle of MultiGroupHead.predict (decode, score filter, rotated NMS,
frustum / direction / range post filters) in numpy float32 + the C rotated-NMS oracle.

Follows, line by line:
  det3d/core/bbox/box_torch_ops.py:81-147       second_box_decode         (pinned: tests/golden/decode_ref.npz)
  det3d/models/bbox_heads/mg_head_sessd.py:893-943   predict              (views / decode)
  det3d/models/bbox_heads/mg_head_sessd.py:945-1057  get_task_detections  (sigmoid, >=0.3, IoU rectification,
                                                                            rotate_nms, frustum, direction, range)
  det3d/core/bbox/box_torch_ops.py:527-548      rotate_nms (topk <= pre_max, keep[:post_max])
  det3d/ops/nms/nms_cpu.py:40-51 + nms_cpu.h:72-168  rotate_nms_cc -> oracle/rotate_nms.c
  det3d/core/bbox/geometry.py:215-277,351-377   points_in_convex_polygon_3d_jit (pinned: nms_helpers_ref.npz)
  det3d/core/bbox/box_np_ops.py:780-833         create_anchors_3d_range   (pinned: nms_helpers_ref.npz)
  det3d/core/bbox/box_np_ops.py:637-654,995-1004,1192-1212  get_valid_frustum chain (pinned: nms_helpers_ref.npz)
Tie rule: torch.topk / numpy argsort leave equal scores unordered; the oracle orders ties by ascending
anchor index (documented in DESIGN.md).
"""
import numpy as np

from . import capi


def create_anchors_3d_range(feature_size=(1, 200, 176), anchor_range=(0, -40.0, -1.0, 70.4, 40.0, -1.0),
                            sizes=(1.6, 3.9, 1.56), rotations=(0, 1.57)):
    """box_np_ops.py:780-833 for one size: returns (D,H,W,1,R,7) float32 [x,y,z,w,l,h,r]."""
    dtype = np.float32
    ar = np.array(anchor_range, dtype)
    stride = (ar[3] - ar[0]) / feature_size[2]
    zc = np.linspace(ar[2], ar[5], feature_size[0], dtype=dtype)
    yc = np.linspace(ar[1], ar[4], feature_size[1], endpoint=False, dtype=dtype) + stride / 2
    xc = np.linspace(ar[0], ar[3], feature_size[2], endpoint=False, dtype=dtype) + stride / 2
    rot = np.array(rotations, dtype)
    D, H, W, R = feature_size[0], feature_size[1], feature_size[2], len(rot)
    out = np.zeros((D, H, W, 1, R, 7), dtype)
    out[..., 0] = xc.reshape(1, 1, W, 1, 1)
    out[..., 1] = yc.reshape(1, H, 1, 1, 1)
    out[..., 2] = zc.reshape(D, 1, 1, 1, 1)
    out[..., 3:6] = np.array(sizes, dtype).reshape(1, 1, 1, 1, 1, 3)
    out[..., 6] = rot.reshape(1, 1, 1, 1, R)
    return out


def second_box_decode(enc, anchors):
    """box_torch_ops.py:81-147 (box_ndim 7, no angle vector, no smooth_dim), float32."""
    enc = np.asarray(enc, np.float32)
    a = np.asarray(anchors, np.float32)
    xa, ya, za, wa, la, ha, ra = [a[..., i] for i in range(7)]
    xt, yt, zt, wt, lt, ht, rt = [enc[..., i] for i in range(7)]
    diag = np.sqrt(la ** 2 + wa ** 2)
    xg = xt * diag + xa
    yg = yt * diag + ya
    zg = zt * ha + za
    lg = np.exp(lt) * la
    wg = np.exp(wt) * wa
    hg = np.exp(ht) * ha
    rg = rt + ra
    return np.stack([xg, yg, zg, wg, lg, hg, rg], -1).astype(np.float32)


# ---- frustum (calibration -> 6 inward-facing planes), float64 like the reference
def projection_matrix_to_CRT_kitti(proj):
    CR = proj[0:3, 0:3]
    CT = proj[0:3, 3]
    RinvCinv = np.linalg.inv(CR)
    Rinv, Cinv = np.linalg.qr(RinvCinv)
    C = np.linalg.inv(Cinv)
    R = np.linalg.inv(Rinv)
    T = Cinv @ CT
    return C, R, T


def get_valid_frustum(rect, Trv2c, P2, image_shape):
    """box_np_ops.py:995-1004: (1,6,4,3) float64 surfaces of the camera frustum in lidar coordinates."""
    C, R, T = projection_matrix_to_CRT_kitti(P2)
    b = [0, 0, image_shape[1], image_shape[0]]
    fku, fkv = C[0, 0], -C[1, 1]
    u0v0 = C[0:2, 2]
    near, far = 0.001, 100
    z = np.array([near] * 4 + [far] * 4, dtype=C.dtype)[:, None]
    bc = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C.dtype)
    nb = (bc - u0v0) / np.array([fku / near, -fkv / near], dtype=C.dtype)
    fb = (bc - u0v0) / np.array([fku / far, -fkv / far], dtype=C.dtype)
    fr = np.concatenate([np.concatenate([nb, fb], 0), z], 1)
    fr = fr - T
    fr = np.linalg.inv(R) @ fr.T
    pts = fr.T
    # camera_to_lidar (box_np_ops.py:637-654)
    pts = np.concatenate([pts, np.ones((pts.shape[0], 1))], -1)
    lidar = pts @ np.linalg.inv((rect @ Trv2c).T)
    corners = lidar[..., :3]
    idx = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)
    return corners[idx][None]


def frustum_planes(surfaces):
    """geometry.py:351-377 surface_equ_3d_jitv2: normal (6,3) and d (6,) of polygon 0, float64."""
    s = np.asarray(surfaces, np.float64)[0]
    sv0 = s[:, 0] - s[:, 1]
    sv1 = s[:, 1] - s[:, 2]
    n = np.cross(sv0, sv1)
    d = -(s[:, 0] * n).sum(-1)
    return n, d


def points_in_frustum(points, surfaces):
    """geometry.py:215-277: inside iff n.p + d < 0 for every plane (float32 points promoted to float64)."""
    n, d = frustum_planes(surfaces)
    p = np.asarray(points, np.float32).astype(np.float64)
    sign = p[:, 0:1] * n[None, :, 0] + p[:, 1:2] * n[None, :, 1] + p[:, 2:3] * n[None, :, 2] + d[None]
    return ~(sign >= 0).any(1)


def sigmoid32(x):
    # clamped at +-88 (exp overflows float32 beyond): sigmoid is 0 / 1 to the last bit there, and numpy stays silent
    x = np.clip(np.asarray(x, np.float32), np.float32(-88.0), np.float32(88.0))
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def predict_frame(box_codes, cls_logits, dir_logits, iou_preds, anchors, frustum, score_thresh=0.3, pre_max=1000,
                  post_max=100, nms_thresh=0.01, post_center_range=(0, -40.0, -5.0, 70.4, 40.0, 5.0),
                  direction_offset=0.0, return_debug=False, forced=None):
    """One frame of get_task_detections. box_codes (A,7), cls_logits (A,), dir_logits (A,2), iou_preds (A,),
    anchors (A,7), frustum (1,6,4,3) float64 or None. Returns dict(box3d_lidar (n,7), scores (n,), label_preds (n,))."""
    boxes = second_box_decode(box_codes, anchors)
    dir_labels = (dir_logits[:, 1] > dir_logits[:, 0]).astype(np.int64)  # torch.max returns the first maximum
    scores = sigmoid32(cls_logits)
    keep = scores >= np.float32(score_thresh)
    idx = np.nonzero(keep)[0]
    # code_scale (ours, for oracle/compare.py): the largest |x / y / z box code| of the frame's head output map -- the scale the
    # float32 error of ANY code of this frame is proportional to (a code is a 128-term dot product over the same feature maps)
    code_scale = float(np.abs(np.asarray(box_codes, np.float32)[:, :3]).max()) if len(box_codes) else 0.0
    dbg = dict(num_candidates=len(idx))
    dbg["code_scale"] = code_scale
    out_empty = dict(box3d_lidar=np.zeros((0, 7), np.float32), scores=np.zeros((0,), np.float32),
                     label_preds=np.zeros((0,), np.int64))
    if len(idx) == 0:
        return (out_empty, dbg) if return_debug else out_empty
    s = scores[idx]
    rect = ((iou_preds[idx].astype(np.float32) + np.float32(1)) * np.float32(0.5)).astype(np.float32)
    s = (s * (rect * rect * rect * rect)).astype(np.float32)  # torch.pow(x, 4)
    b = boxes[idx]
    dl = dir_labels[idx]
    # rotate_nms: topk(min(K, pre_max)) then rotate_nms_cc, keep[:post_max]
    k = min(len(idx), pre_max)
    order = np.lexsort((idx, -s.astype(np.float64)))[:k]  # score desc, ties by anchor index asc
    dets = np.concatenate([b[order][:, [0, 1, 3, 4, 6]], s[order][:, None]], 1).astype(np.float32)
    kept, near, pairs = capi.rotate_nms_cc(dets, nms_thresh, order=np.arange(k, dtype=np.int32), return_pairs=True, forced=forced)
    kept = kept[:post_max]
    sel = order[kept]
    # the NMS problem itself, for oracle/compare.py: candidates in score order, who survived, which decisions were marginal
    dbg.update(cand_dets=dets, cand_boxes=b[order], cand_anchor=idx[order], nms_kept_rows=kept, near_pairs=pairs)
    b, s, dl = b[sel], s[sel], dl[sel]
    dbg.update(topk=k, nms_kept=len(sel), near_threshold_pairs=near, selected_anchor=idx[sel])
    if frustum is not None and len(b):
        m = points_in_frustum(b[:, :3], frustum)
        b, s, dl = b[m], s[m], dl[m]
        dbg["selected_anchor"] = dbg["selected_anchor"][m]
    if len(b) == 0:
        return (out_empty, dbg) if return_debug else out_empty
    opp = ((b[:, 6] - np.float32(direction_offset)) > 0) ^ (dl == 1)
    b = b.copy()
    b[:, 6] = b[:, 6] + np.where(opp, np.float32(np.pi), np.float32(0.0)).astype(np.float32)
    pr = np.array(post_center_range, np.float32)
    m = (b[:, :3] >= pr[:3]).all(1) & (b[:, :3] <= pr[3:]).all(1)
    dbg["selected_anchor"] = dbg["selected_anchor"][m]
    out = dict(box3d_lidar=b[m], scores=s[m], label_preds=np.zeros((int(m.sum()),), np.int64))
    return (out, dbg) if return_debug else out
