"""Host-side pieces of the training loop around the captured iteration (sessd_hip/trainloop.py; reference trainer_sessd.py:306-360):
the synthetic labels agree with the scan geometry, the held-out scans form a KITTI-format validation set whose own boxes score
AP 100 and whose perturbed boxes do not, the consistency ramp-up. The rotated overlaps of the evaluation are served by the CPU
oracle here (the product takes them from the device kernel)."""
import numpy as np

from oracle import capi
from sessd_hip import synth, trainloop


def _oracle_rotate_iou(boxes, query_boxes, criterion=-1, device_id=0):
    if boxes.shape[0] == 0 or query_boxes.shape[0] == 0:
        return np.zeros((boxes.shape[0], query_boxes.shape[0]), np.float32)
    return capi.rotate_iou_eval(boxes.astype(np.float32), query_boxes.astype(np.float32), criterion).astype(boxes.dtype)


def test_labels_sit_on_the_points_they_label():
    """frame_cars' yaw in the det3d lidar-box convention (length axis = (sin r, cos r)): the box grown by 10 cm holds nearly all
    points within 2.2 m of a visible car's centre above the ground; the box a quarter turn off (rounds 1 - 4) holds far fewer."""
    from det3d.core.bbox import box_np_ops
    inside = outside = wrong = 0
    for seed in (3, 4, 5):
        pts = synth.make_frame(seed, None)
        cars, counts = synth.frame_labels(seed, pts)
        assert np.all(np.abs(cars[:, 6]) <= np.pi + 1e-6)
        vis = cars[counts >= 50]
        assert len(vis) >= 2
        above = pts[pts[:, 2] > -1.6]
        for c in vis:
            near = above[np.linalg.norm(above[:, :2] - c[:2], axis=1) < 1.6]   # closer than any other object can be placed ... mostly
            g = c[None].copy(); g[:, 3:6] += 0.1
            inside += int(box_np_ops.points_in_rbbox(near[:, :3], g, z_axis=2, origin=(0.5, 0.5, 0.5)).sum())
            outside += len(near)
            q = g.copy(); q[:, 6] = np.pi / 2 - q[:, 6]
            wrong += int(box_np_ops.points_in_rbbox(near[:, :3], q, z_axis=2, origin=(0.5, 0.5, 0.5)).sum())
    assert inside > 0.9 * outside and wrong < 0.8 * inside, (inside, outside, wrong)


def test_synthetic_validation_set_scores_its_own_boxes(monkeypatch):
    import det3d.datasets.utils.eval as U
    monkeypatch.setattr(U, "rotate_iou_gpu_eval", _oracle_rotate_iou)
    pool = trainloop.ScenePool(range(900, 912), 20000)
    val = trainloop.SyntheticKitti(pool)
    n_gt = sum(int((np.asarray(i["annos"]["occluded"]) == 0).sum()) for i in val.infos)
    assert n_gt >= 30 and all(set(i["annos"]) >= {"name", "bbox", "location", "dimensions", "rotation_y", "occluded"} for i in val.infos)
    rng = np.random.RandomState(0)
    def dets(noise, drop):
        out = []
        for i in range(len(pool)):
            b = pool.visible(i).copy()
            keep = rng.rand(len(b)) >= drop
            b = b[keep]
            b[:, :3] += rng.normal(0, noise, (len(b), 3)) + 0.01   # (identical rotated boxes are ill-conditioned in the overlap)
            out.append(dict(box3d_lidar=b, scores=rng.uniform(0.5, 1.0, len(b)).astype(np.float32), label_preds=np.zeros(len(b), np.int64)))
        return out
    perfect = val.evaluate(dets(0.0, 0.0))
    # (with a few dozen objects not every one of the 11 / 41 recall samples is reached: AP = reached samples / 11, eval.py
    # get_thresholds -- precision is 1 at every reached one)
    assert min(perfect["ap3d_11"]) > 90.0 and min(perfect["ap3d_40"]) > 90.0 and perfect["gt_cars"] == n_gt
    print(perfect["ap3d_11"], perfect["ap3d_40"])
    worse = val.evaluate(dets(0.25, 0.3))
    print(worse["ap3d_11"], worse["ap3d_40"])
    assert max(worse["ap3d_11"]) < 80.0 and worse["ap3d_11"][1] < perfect["ap3d_11"][1] - 15
    # a detection ON a car with too few points to be ground truth is ignored, not a false positive
    extra = dets(0.0, 0.0)
    added = 0
    for i in range(len(pool)):
        few = pool.cars[i][(pool.counts[i] > 0) & (pool.counts[i] < trainloop.MIN_POINTS)]
        if len(few):
            e = extra[i]
            e["box3d_lidar"] = np.concatenate([e["box3d_lidar"], few + 0.01])
            e["scores"] = np.concatenate([e["scores"], np.full(len(few), 0.99, np.float32)])
            e["label_preds"] = np.zeros(len(e["scores"]), np.int64)
            added += len(few)
    got = val.evaluate(extra)
    assert added > 0 and got["ap3d_11"] == perfect["ap3d_11"] and got["ap3d_40"] == perfect["ap3d_40"]


def test_consistency_ramp_up():
    """trainer_sessd.py:306-312: exp(-5 (1 - e/15)^2) over the first 15 of 60 epochs, then 1"""
    w = [trainloop.consistency_weight(i, 6000) for i in (0, 750, 1499, 1500, 5999)]
    assert abs(w[0] - np.exp(-5.0)) < 1e-9 and w[0] < w[1] < w[2] < 1.0 and w[3] == 1.0 and w[4] == 1.0
