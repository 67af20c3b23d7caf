"""TEST INFRASTRUCTURE ONLY -- the whole reference inference path on the CPU, stage by stage:
voxelize (oracle/voxelize.c) -> mean VFE -> SpMiddleFHD (oracle/sparse_conv.py) -> SSFA + heads
(oracle/dense_head.py) -> predict (oracle/postprocess.py + oracle/rotate_nms.c).
Mirrors VoxelNet.forward(example, return_loss=False) (det3d/models/detectors/voxelnet_sessd.py:18-43).
Also the `cpu_baseline` of bench.py ("port": the reference itself cannot run here -- SURVEY.md 8c)."""
import time

import numpy as np
import torch

from . import capi, dense_head, postprocess, sparse_conv


def split_state_dict(sd):
    convs, bns = [], []
    for i in range(14):
        convs.append(sd["backbone.middle_conv.%d.weight" % (3 * i)].float())
        p = "backbone.middle_conv.%d." % (3 * i + 1)
        bns.append({k: sd[p + k].float() for k in ("weight", "bias", "running_mean", "running_var")})
    return convs, bns


def run_frames(points_list, sd, voxel_range, voxel_size, max_points, max_voxels, anchors, frustum=None, test_cfg=None,
               timings=None, return_intermediate=False):
    """points_list: list of (P,4) float32 numpy. Returns list of per-frame detection dicts (numpy)."""
    tc = dict(score_thresh=0.3, pre_max=1000, post_max=100, nms_thresh=0.01)
    if test_cfg:
        tc.update(test_cfg)
    T = timings if timings is not None else {}

    def tick(name, t0):
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0

    t0 = time.perf_counter()
    feats, coors = [], []
    grid = np.round((np.array(voxel_range[3:], np.float32) - np.array(voxel_range[:3], np.float32)) / np.array(voxel_size, np.float32)).astype(np.int64)
    for b, pts in enumerate(points_list):
        v, c, n = capi.points_to_voxel(pts, voxel_size, voxel_range, max_points, max_voxels)
        feats.append(capi.vfe_mean(v, n, 4))
        coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    feats = torch.from_numpy(np.concatenate(feats, 0))
    coors = np.concatenate(coors, 0)
    tick("voxelize", t0)
    t0 = time.perf_counter()
    convs, bns = split_state_dict(sd)
    B = len(points_list)
    bev, levels = sparse_conv.spmiddle_fhd(feats, coors, B, [int(g) for g in grid], convs, bns, return_levels=True)
    tick("spmiddle", t0)
    t0 = time.perf_counter()
    x = dense_head.ssfa_forward(bev, sd)
    preds = dense_head.head_forward(x, sd)
    tick("ssfa_head", t0)
    t0 = time.perf_counter()
    out = []
    dbg = []
    for b in range(B):
        box = preds["box_preds"][b].reshape(-1, 7).numpy()
        cls = preds["cls_preds"][b].reshape(-1).numpy()
        dirl = preds["dir_cls_preds"][b].reshape(-1, 2).numpy()
        iou = preds["iou_preds"][b].reshape(-1).numpy()
        fr = None if frustum is None else frustum[b]
        args = (box, cls, dirl, iou, anchors, fr, tc["score_thresh"], tc["pre_max"], tc["post_max"], tc["nms_thresh"])
        r, d = postprocess.predict_frame(*args, return_debug=True)
        # the same frame with chosen near-threshold NMS decisions taken the other way (oracle/compare.py)
        d["rerun"] = (lambda a: (lambda forced: postprocess.predict_frame(*a, forced=forced)))(args)
        out.append(r)
        dbg.append(d)
    tick("predict", t0)
    if return_intermediate:
        return out, dict(bev=bev, levels=levels, ssfa=x, preds=preds, debug=dbg, num_voxels=coors.shape[0])
    return out
