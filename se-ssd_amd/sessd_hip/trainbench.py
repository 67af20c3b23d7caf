"""BASELINE configs[2] ("Single MI355X SE-SSD training step: teacher + student forward, ODIoU + consistency loss, EMA, batch 4")
as a measurable unit: a labelled synthetic batch in the reference's collated-example form and the timing of the captured
iteration. Used by bench.py (the `train_step` key of the driver's line) and scripts/train_step_bench.py.

The batch (reference data flow: pipelines/preprocess.py:31-175 -> Voxelization :196-232 -> AssignTarget :236-358 -> Reformat ->
collate_kitti): per sample a synthetic scan (sessd_hip.synth.make_frame) with its 15 car boxes as ground truth; the TEACHER sees
the raw cloud (`*_raw` keys), the STUDENT the globally augmented one (flip about the x axis, rotation about z, scaling -- recorded
in `transformation`, which the consistency loss uses to map the teacher's boxes into the student's frame, mg_head_sessd.py:670-674);
targets by the device AssignTarget kernel (sessd_assign_targets) against the 70400 anchors."""
import math
import time

import numpy as np
import torch

from . import configs, ops, synth
from . import train as strain

VG = configs.VOXEL_GENERATOR


def _augment(points, boxes, flipped, rot, scale):
    """Global augmentation of preprocess.py:137-140 (random_flip / global_rotation / global_scaling_v2) with the recorded values."""
    p, b = points.copy(), boxes.copy()
    if flipped:
        p[:, 1] = -p[:, 1]
        b[:, 1] = -b[:, 1]
        b[:, 6] = -b[:, 6] + np.pi
    c, s = math.cos(rot), math.sin(rot)
    for a in (p, b):  # points @ [[c, -s, 0], [s, c, 0], [0, 0, 1]] (box_np_ops.rotation_points_single_angle, axis 2)
        x, y = a[:, 0].copy(), a[:, 1].copy()
        a[:, 0], a[:, 1] = x * c + y * s, -x * s + y * c
    b[:, 6] += rot
    p[:, :3] *= scale
    b[:, :6] *= scale
    return p.astype(np.float32), b.astype(np.float32)


def labelled_batch(dev, batch=4, seed0=50, npts=20000, max_voxels=16000, aug_seed=0):
    """(example, capacity-form example) of `batch` labelled synthetic samples, everything on `dev`."""
    from .anchors import create_anchors_3d_range
    rng = np.random.RandomState(aug_seed)
    anchors = torch.from_numpy(create_anchors_3d_range((1, 200, 176)).reshape(-1, 7).astype(np.float32)).to(dev)
    A = anchors.shape[0]
    raw_pts, stu_pts, trans, lab, reg, lab_raw, reg_raw = [], [], [], [], [], [], []
    lo, hi = np.array(VG["range"][:2]), np.array(VG["range"][3:5])
    for i in range(batch):
        pts, cars = synth.make_frame(seed0 + i, npts), synth.frame_cars(seed0 + i)
        t = dict(flipped=bool(rng.rand() < 0.5), noise_rotation=float(rng.uniform(-math.pi / 4, math.pi / 4)),
                 noise_scale=float(rng.uniform(0.95, 1.05)))
        p2, cars2 = _augment(pts, cars, t["flipped"], t["noise_rotation"], t["noise_scale"])
        for boxes, L, R in ((cars2, lab, reg), (cars, lab_raw, reg_raw)):
            keep = np.all((boxes[:, :2] >= lo) & (boxes[:, :2] <= hi), 1)   # Voxelization's ground-truth range filter
            tg = ops.assign_targets(anchors, torch.from_numpy(boxes[keep]).to(dev), None, 0.6, 0.45)
            L.append(tg["labels"])
            R.append(tg["bbox_targets"])
        raw_pts.append(torch.from_numpy(pts).to(dev))
        stu_pts.append(torch.from_numpy(p2).to(dev))
        trans.append(t)
    ex = dict(shape=[[1408, 1600, 40]] * batch, metadata=[{}] * batch, transformation=trans)
    for suffix, clouds in (("", stu_pts), ("_raw", raw_pts)):
        r = ops.voxelize_batch(clouds, VG["voxel_size"], VG["range"], 5, max_voxels)
        pre = r["prefix"].cpu().numpy()
        m = int(pre[batch])
        ex["voxels" + suffix], ex["coordinates" + suffix], ex["num_points" + suffix] = r["voxels"][:m], r["coors"][:m], r["num_points"][:m]
        ex["num_voxels" + suffix] = torch.tensor(np.diff(pre))
        ex["shape" + suffix] = ex["shape"]
    ab = anchors.unsqueeze(0).repeat(batch, 1, 1).contiguous()
    ex.update(anchors=[ab], anchors_raw=[ab.clone()], labels=[torch.stack(lab).contiguous()], reg_targets=[torch.stack(reg).contiguous()],
              labels_raw=[torch.stack(lab_raw).contiguous()], reg_targets_raw=[torch.stack(reg_raw).contiguous()])
    m = max(int(ex["voxels"].shape[0]), int(ex["voxels_raw"].shape[0]))
    cap = strain.capacity_example(ex, (int(m * 1.08) + 4095) // 4096 * 4096)   # 8 % headroom over this batch's voxels
    return ex, cap


def measure(dev, batch=4, steps=20, warmup=3, real_loss=True, standin_loss_fn=None, seed=0, pretrain=300, scenes=24):
    """Capture the whole iteration as ONE hipGraph (teacher forward, student forward, loss, backward, flat all-reduce hook,
    fused clip + Adam + EMA) on a labelled synthetic batch and time `steps` replays. real_loss=True: the reference loss
    (MultiGroupHead.loss + consistency loss through sessd_head_loss) -- BASELINE configs[2]; False: `standin_loss_fn`
    (round 3's slice). Returns a dict.
    pretrain (round 5, real loss only): the timed replays come AFTER `pretrain` captured iterations on fresh batches of a small
    scene pool (sessd_hip.trainloop: device data path), so that the timed iteration exercises what a training run exercises --
    teacher and student agree on boxes (matched pairs, consistency loss > 0; the round-4 figure was taken at iteration 3 of a
    random network: `matched_boxes` 0). `ms_per_iter` stays the replay of one resident batch (comparable with earlier rounds);
    `ms_per_iter_fresh_batches` has the data path inside the clock."""
    model = configs.build_synthetic_detector(dev, seed=seed)
    fresh = None
    if real_loss and pretrain > 0:
        from . import trainloop
        pool = trainloop.ScenePool(range(50, 50 + scenes), 20000)
        step = strain.TrainStep(model, None, total_steps=2 * pretrain + 100)
        data = trainloop.DeviceBatcher(pool, dev, batch, pretrain + 2, seed=seed)
        cap = ex = data.load(0)
        step.capture(cap, consistency_weight=1.0, warmup=max(1, warmup))
        side, main = torch.cuda.Stream(device=dev), torch.cuda.current_stream()
        data.load(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(1, pretrain + 1):   # trainloop.fit's loop: the next batch is assembled on a side stream while this one trains
            free = main.record_event()
            step.replay()
            side.wait_event(free)
            with torch.cuda.stream(side):
                data.load(it + 1, into=data.staging())
            main.wait_stream(side)
            data.commit()
        torch.cuda.synchronize()
        fresh = (time.perf_counter() - t0) / pretrain * 1e3
    else:
        step = strain.TrainStep(model, None if real_loss else standin_loss_fn, total_steps=1000)
        ex, cap = labelled_batch(dev, batch)
        step.capture(cap, consistency_weight=1.0, warmup=max(1, warmup))
    for _ in range(3):
        step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"what": "SE-SSD training iteration as ONE captured hipGraph: teacher forward (raw cloud) + student forward (augmented "
                   "cloud) + %s + backward + fused clip / Adam / EMA update; batch %d x 20 k-point synthetic scans with their cars as "
                   "ground truth, targets by sessd_assign_targets, random-init weights"
                   % ("MultiGroupHead.loss (focal + ODIoU + direction + IoU-prediction) + teacher-student consistency loss as the "
                      "capacity-form device op sessd_head_loss" if real_loss else "a stand-in loss on the head outputs", batch),
           "config": "BASELINE.json configs[2]" if real_loss else "slice (stand-in loss)", "batch": batch, "replays": steps,
           "ms_per_iter": ms, "samples_per_s": batch / ms * 1e3,
           "voxels_student": int(ex["num_voxels_dev"].item()) if "num_voxels_dev" in ex else int(ex["voxels"].shape[0]),
           "voxels_teacher": int(ex["num_voxels_dev_raw"].item()) if "num_voxels_dev_raw" in ex else int(ex["voxels_raw"].shape[0]),
           "sparse_overflow_flag": int(step.sparse_overflow.item()) if step.sparse_overflow is not None else 0,
           "loss": float(step.static_loss)}
    if fresh is not None:
        out["pretrain_iterations"] = pretrain
        out["ms_per_iter_fresh_batches"] = fresh
        out["samples_per_s_fresh_batches"] = batch / fresh * 1e3
        out["what"] += "; timed after %d captured iterations on fresh batches of %d scenes (visible cars as ground truth)" % (pretrain, scenes)
    if real_loss:
        L = step.student.bbox_head.record_to_dict(step.last_record)
        R = ops.HEAD_LOSS_RECORD
        rec = step.last_record.cpu()
        out["loss_terms"] = {k: float(L[k][0].sum()) for k in ("loss", "cls_loss_reduced", "ious_loss", "dir_loss_reduced", "iou_pred_loss",
                                                              "consistency_loss", "loss_ema")}
        out["positives"], out["consistency_candidates"] = int(rec[R["positives"]]), [int(rec[R["candidates"]]), int(rec[R["candidates_ema"]])]
        out["matched_boxes"], out["loss_overflow_flags"] = int(rec[R["matched_boxes"]]), int(rec[R["overflow"]])
    return out, step
