"""The training LOOP around the captured SE-SSD iteration (reference: det3d/torchie/trainer/trainer_sessd.py:250-275 the
iteration, :306-312 the consistency ramp-up, :315-360 the epoch loop with the EMA update; data flow pipelines/preprocess.py:31-175
-> Voxelization :196-232 -> AssignTarget :236-358 -> collate) on the only labelled data that exists in this environment: the
ray-cast synthetic scans of sessd_hip.synth with their VISIBLE cars as ground truth.

  ScenePool      a set of scans + labels, generated once (host, numpy; optionally on several processes)
  DeviceBatcher  a fresh labelled batch per iteration, assembled ON THE DEVICE inside the static capacity-form example a captured
                 TrainStep replays on: the clouds are resident, the student's global augmentation (random flip about the x axis,
                 rotation about z in +-pi/4, scaling in 0.95 .. 1.05: config.py:163-166, preprocess.py:137-140) is applied to
                 points and boxes by device arithmetic on per-iteration parameters that were uploaded once, both networks' inputs
                 are voxelized by sessd_voxelize_frames (teacher: the raw cloud), targets come from sessd_assign_targets, the
                 recorded augmentation goes to `transformation_dev`. Nothing reads the host, so the host runs ahead of the device
                 and the data path overlaps the previous iteration.
  fit            the loop: load a batch, replay the graph, every `log_every` iterations ONE host read of the 64-float loss record
                 (and of the sticky overflow flags); consistency weight by the reference's sigmoid ramp-up over the first quarter
                 of the run (15 of 60 epochs); returns the log and the sustained samples/s (data path INSIDE the clock).
  SyntheticKitti the held-out scans as a KITTI-format validation set: kitti_infos with camera-frame annotations made by the very
                 conversion detections go through (det3d.datasets.kitti.convert_detection_to_kitti_annos), for KittiDataset.evaluation.
The GT-AUG database sampler and the per-object noise of the reference's Preprocess stay with the host mirror
(det3d/datasets/pipelines/preprocess.py); this loop uses the global augmentation only -- the part the consistency loss depends on."""
import math
import time

import numpy as np
import torch

from . import configs, ops, synth
from . import train as strain

VG = configs.VOXEL_GENERATOR
MIN_POINTS = 8     # a car with at least this many points of the scan on it is ground truth
FAR = -1000.0      # where a box that left the range is parked: it overlaps no anchor, so create_target_np ignores it


def _scene(args):
    seed, npts = args
    pts = synth.make_frame(seed, npts)
    cars, counts = synth.frame_labels(seed, pts)
    return seed, pts, cars, counts


class ScenePool:
    """`seeds` -> frames (P, 4) float32, the 15 placed cars (15, 7) and how many points each carries."""

    def __init__(self, seeds, npts=20000, workers=0):
        seeds = [int(s) for s in seeds]
        jobs = [(s, npts) for s in seeds]
        if workers and workers > 1 and len(jobs) > 4:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(workers) as pool:   # spawn: the parent may already hold a HIP context
                rows = pool.map(_scene, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
        else:
            rows = [_scene(j) for j in jobs]
        self.seeds = [r[0] for r in rows]
        self.frames = [r[1] for r in rows]
        self.cars = [r[2] for r in rows]
        self.counts = [r[3] for r in rows]

    def __len__(self):
        return len(self.frames)

    def visible(self, i, min_points=MIN_POINTS):
        return self.cars[i][self.counts[i] >= min_points]


class DeviceBatcher:
    """Labelled batches of `batch` scenes in ONE static capacity-form example (see the module docstring). `iterations` fixes how
    many batches can be drawn: their scene choices and augmentation parameters are drawn here, once, from `seed`."""

    M = 15  # boxes per scene (visible ones first; the rest parked at FAR)

    def __init__(self, pool, dev, batch=4, iterations=1000, max_voxels=16000, seed=0, augment=True):
        from .anchors import create_anchors_3d_range
        self.pool, self.dev, self.B, self.max_voxels = pool, dev, int(batch), int(max_voxels)
        self.T = int(iterations)
        rng = np.random.RandomState(seed)
        B, T, M = self.B, self.T, self.M
        self.choice = np.stack([rng.choice(len(pool), B, replace=len(pool) < B) for _ in range(T)])
        flip = (rng.rand(T, B) < 0.5) if augment else np.zeros((T, B), bool)
        rot = rng.uniform(-math.pi / 4, math.pi / 4, (T, B)) if augment else np.zeros((T, B))
        scale = rng.uniform(0.95, 1.05, (T, B)) if augment else np.ones((T, B))
        # [flipped, cos, sin, rotation, scale]: the row layout of MultiGroupHead.transformation_tensor
        par = np.stack([flip.astype(np.float64), np.cos(rot), np.sin(rot), rot, scale], -1).astype(np.float32)
        self.par_host = par
        self.par = torch.from_numpy(par).to(dev)
        self.frames = [torch.from_numpy(f).to(dev) for f in pool.frames]
        boxes = np.full((len(pool), M, 7), 1.0, np.float32)
        boxes[:, :, 0] = FAR
        for i in range(len(pool)):
            v = pool.visible(i)
            boxes[i, :len(v)] = v
        self.boxes = torch.from_numpy(boxes).to(dev)
        self.anchors = torch.from_numpy(create_anchors_3d_range((1, 200, 176)).reshape(-1, 7).astype(np.float32)).to(dev)
        self.lo = torch.tensor(VG["range"][:2], dtype=torch.float32, device=dev)
        self.hi = torch.tensor(VG["range"][3:5], dtype=torch.float32, device=dev)
        self.cap = B * self.max_voxels
        self.rows = torch.arange(self.cap, device=dev, dtype=torch.int32)
        A = self.anchors.shape[0]
        i32, f32 = torch.int32, torch.float32
        Z = lambda *s, dt=f32: torch.zeros(s, dtype=dt, device=dev)
        ab = self.anchors.unsqueeze(0).repeat(B, 1, 1).contiguous()
        ex = dict(shape=[[1408, 1600, 40]] * B, shape_raw=[[1408, 1600, 40]] * B, metadata=[{}] * B,
                  num_voxels=torch.zeros(B, dtype=torch.int64), num_voxels_raw=torch.zeros(B, dtype=torch.int64),
                  anchors=[ab], anchors_raw=[ab.clone()], labels=[Z(B, A, dt=i32)], reg_targets=[Z(B, A, 7)],
                  labels_raw=[Z(B, A, dt=i32)], reg_targets_raw=[Z(B, A, 7)], transformation_dev=Z(B, 5))
        for sfx in ("", "_raw"):
            ex["voxels" + sfx] = Z(self.cap, 5, 4)
            ex["coordinates" + sfx] = torch.full((self.cap, 4), -1, dtype=i32, device=dev)
            ex["num_points" + sfx] = torch.ones(self.cap, dtype=i32, device=dev)
            ex["num_voxels_dev" + sfx] = Z(1, dt=i32)
        self.example = ex
        self.cursor = 0

    # ---- device arithmetic of the global augmentation (preprocess.py:137-140 / box_np_ops.rotation_points_single_angle, axis 2)
    @staticmethod
    def _move(xy_x, xy_y, p):
        """flip about the x axis, then rotate by p[3] (points @ [[c, -s], [s, c]]), then scale -- on (x, y) columns"""
        y = torch.where(p[0] != 0, -xy_y, xy_y)
        c, s = p[1], p[2]
        return (xy_x * c + y * s) * p[4], (-xy_x * s + y * c) * p[4]

    def _student_cloud(self, pts, p):
        out = torch.empty_like(pts)
        x, y = self._move(pts[:, 0], pts[:, 1], p)
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = x, y, pts[:, 2] * p[4], pts[:, 3]
        return out

    def _student_boxes(self, b, p):
        out = torch.empty_like(b)
        x, y = self._move(b[:, 0], b[:, 1], p)
        out[:, 0], out[:, 1] = x, y
        out[:, 2:6] = b[:, 2:6] * p[4]
        r = torch.where(p[0] != 0, -b[:, 6] + math.pi, b[:, 6])
        out[:, 6] = r + p[3]
        return out

    def _in_range(self, b):
        """Voxelization's ground-truth range filter (preprocess.py:205-210 -> core/sampler/preprocess.py:138-148
        filter_gt_box_outside_range) without a host-read shape: a box is kept when ANY of its four BEV corners (dims (w, l) about
        the centre, rotated by box_np_ops.rotation_2d's matrix: x' = x c + y s, y' = -x s + y c) lies STRICTLY inside the range
        -- the reference's cross-product test fails on the boundary --, not only its centre (that is the reference's OTHER
        function, filter_gt_box_outside_range_by_center; round-5 advisor finding). A dropped box is parked where it overlaps no
        anchor (create_target_np then ignores it: empty_gt_mask, target_ops_v3.py:62-66)."""
        c, s = torch.cos(b[:, 6]), torch.sin(b[:, 6])
        keep = torch.zeros(b.shape[0], dtype=torch.bool, device=b.device)
        for fx, fy in ((-0.5, -0.5), (-0.5, 0.5), (0.5, 0.5), (0.5, -0.5)):
            dx, dy = fx * b[:, 3], fy * b[:, 4]
            x, y = dx * c + dy * s + b[:, 0], -dx * s + dy * c + b[:, 1]
            keep |= (x > self.lo[0]) & (x < self.hi[0]) & (y > self.lo[1]) & (y < self.hi[1])
        keep &= b[:, 0] > FAR / 2
        far = b.clone()
        far[:, 0] = FAR
        return torch.where(keep[:, None], b, far)

    _MOVING = ("voxels", "coordinates", "num_points", "num_voxels_dev", "voxels_raw", "coordinates_raw", "num_points_raw",
               "num_voxels_dev_raw", "transformation_dev")
    _LISTED = ("labels", "reg_targets", "labels_raw", "reg_targets_raw")

    def staging(self):
        """A second set of the tensors load() writes (made on first use): load(it, into=staging()) on a side stream prepares the
        next batch while the captured iteration runs on the static example, commit() then copies it over (fit(overlap=True))."""
        if getattr(self, "_staging", None) is None:
            st = {k: torch.empty_like(self.example[k]) for k in self._MOVING}
            st.update({k: [torch.empty_like(self.example[k][0])] for k in self._LISTED})
            self._staging = st
            self._pairs = ([self.example[k] for k in self._MOVING] + [self.example[k][0] for k in self._LISTED],
                           [st[k] for k in self._MOVING] + [st[k][0] for k in self._LISTED])
        return self._staging

    def commit(self):
        """staging -> the static example (one multi-tensor copy on the current stream, ~20 MB)"""
        self.staging()
        torch._foreach_copy_(self._pairs[0], self._pairs[1])

    def load(self, it=None, into=None):
        """Fill the static example (or `into`: staging()) with batch `it` (default: the next one). No host synchronisation."""
        it = self.cursor if it is None else int(it)
        if it >= self.T:
            raise IndexError("DeviceBatcher was built for %d iterations" % self.T)
        self.cursor = it + 1
        ex, B = (self.example if into is None else into), self.B
        idx = self.choice[it]
        raw = [self.frames[i] for i in idx]
        stu = [self._student_cloud(self.frames[i], self.par[it, b]) for b, i in enumerate(idx)]
        for sfx, clouds in (("", stu), ("_raw", raw)):
            r = ops.voxelize_frames(clouds, VG["voxel_size"], VG["range"], 5, self.max_voxels)
            n = r["prefix"][B:B + 1]
            live = self.rows < n
            ex["voxels" + sfx].copy_(torch.where(live[:, None, None], r["voxels"], 0.0))
            ex["coordinates" + sfx].copy_(torch.where(live[:, None], r["coors"], -1))
            ex["num_points" + sfx].copy_(torch.where(live, r["num_points"], 1))
            ex["num_voxels_dev" + sfx].copy_(n)
        for b, i in enumerate(idx):
            gt_raw = self._in_range(self.boxes[i])
            gt_stu = self._in_range(self._student_boxes(self.boxes[i], self.par[it, b]))
            for gt, L, R in ((gt_stu, "labels", "reg_targets"), (gt_raw, "labels_raw", "reg_targets_raw")):
                tg = ops.assign_targets(self.anchors, gt, None, 0.6, 0.45)
                ex[L][0][b].copy_(tg["labels"])
                ex[R][0][b].copy_(tg["bbox_targets"])
        ex["transformation_dev"].copy_(self.par[it])
        return ex


def consistency_weight(it, total):
    """trainer_sessd.py:306-312 with the run's `total` iterations standing for the reference's 60 epochs"""
    return strain.consistency_rampup(int(it * 60 // max(1, total)), 60)


def fit(model, pool, iterations=2000, batch=4, lr_max=3e-3, seed=0, log_every=50, capture=True, on_log=None, ema_check=False,
        overlap=True, raise_on_overflow=True):
    """Train `model` (student; the teacher is its EMA copy) for `iterations` captured iterations on fresh batches from `pool`.
    Returns (TrainStep, report). report: log rows (iteration, the record's terms averaged over the window's LAST iteration -- one
    host read per `log_every` iterations), sustained samples/s with the data path inside the clock, overflow flags seen.
    ema_check: also carry teacher_ref = alpha * teacher_ref + (1 - alpha) * student (trainer_sessd.py:315-318) in torch on the
    device after every iteration and report its largest difference from the fused update's teacher.
    overlap: batch i + 1 is assembled on a SIDE stream into staging tensors while iteration i runs, then copied into the static
    example (the reference overlaps its DataLoader workers with the iteration the same way); False: load, then iterate, on one
    stream. Same batches, same arithmetic: the trained parameters are bit-identical either way (tests/test_trainloop_gpu.py).
    raise_on_overflow: with every log record the STICKY device flags are read -- sparse level capacities of both networks' passes
    (TrainStep.sparse_overflow) and the loss capacities (loss_overflow) -- and a set flag raises: the window trained on truncated
    tensors. False only records them in the rows / the report (`sparse_overflow_flag` = OR over the whole run, bit 0 student,
    bit 1 teacher)."""
    dev = next(model.parameters()).device
    step = strain.TrainStep(model, None, total_steps=iterations, lr_max=lr_max)
    data = DeviceBatcher(pool, dev, batch, iterations, seed=seed)
    ex = data.load(0)
    side = torch.cuda.Stream(device=dev) if overlap else None
    warm = 1
    if capture:
        step.capture(ex, consistency_weight=consistency_weight(0, iterations), warmup=warm)   # `warm` real iterations on batch 0
    done = warm if capture else 0
    ema_ref = step.flat_t.data.clone() if ema_check else None
    R = ops.HEAD_LOSS_RECORD
    keys = ("total", "loss", "cls_loss_reduced", "ious_loss", "dir_loss_reduced", "iou_pred_loss", "consistency_loss", "loss_ema",
            "num_pos", "positives", "matched_boxes", "candidates", "candidates_ema", "overflow")
    log, flags, sparse_flags = [], 0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if overlap and done < iterations:
        data.load(done)
    for it in range(done, iterations):
        main = torch.cuda.current_stream()
        if not overlap:
            data.load(it)
        else:
            free = main.record_event()   # the staging set is free from here on: the copy out of it (previous commit) is older
        w = consistency_weight(it, iterations)
        # the iteration is enqueued FIRST (one graph launch): the ~2 ms of host work that assembles the next batch then run while
        # the device is already busy (loader first: 15.1 ms per iteration; iteration first: see profiles/r5_trainloop_probe.json)
        if capture:
            step.replay(consistency_weight=w)
        else:
            step(ex, consistency_weight=w, device_schedule=True)
        if overlap and it + 1 < iterations:
            side.wait_event(free)
            with torch.cuda.stream(side):
                data.load(it + 1, into=data.staging())
            main.wait_stream(side)
            data.commit()
        if ema_ref is not None:
            a = strain.ema_alpha(step.global_step - 1)
            ema_ref.mul_(a).add_(step.flat_s.data, alpha=1.0 - a)
        if (it + 1) % log_every == 0 or it + 1 == iterations:
            rec = step.last_record.detach().cpu().numpy()   # the window's one host read
            row = {"iteration": it + 1, "consistency_weight": w}
            row.update({k: float(rec[R[k]]) for k in keys})
            sticky = int(step.loss_overflow.item()) if step.loss_overflow is not None else 0
            if sticky:
                flags |= sticky
                step.loss_overflow.zero_()
            row["overflow_since_last_log"] = sticky
            sp = int(step.sparse_overflow.item()) if step.sparse_overflow is not None else 0
            if sp:
                sparse_flags |= sp
                step.sparse_overflow.zero_()
            row["sparse_overflow_since_last_log"] = sp
            log.append(row)
            if raise_on_overflow and (sp or sticky):
                raise RuntimeError("capacity overflow in iterations %d .. %d: sparse levels mask %d (bit 0 student, bit 1 teacher pass), "
                                   "loss capacities mask %d (bit 0 positives, bit 1 consistency candidates)"
                                   % (it + 2 - log_every, it + 1, sp, sticky))
            if on_log is not None:
                on_log(row)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = iterations - done
    rep = {"iterations": iterations, "batch": batch, "timed_iterations": n, "seconds": dt, "ms_per_iteration": dt / max(1, n) * 1e3,
           "samples_per_s": n * batch / dt if n else 0.0, "log": log, "overflow_flags": flags, "scenes": len(pool),
           "sparse_overflow_flag": sparse_flags, "data_path_overlapped": bool(overlap),
           "what": "captured SE-SSD iterations (teacher + student forward, reference loss, backward, clip / Adam / EMA) on FRESH "
                   "batches: scene choice + global augmentation + voxelization of both clouds + target assignment on the device "
                   "inside the clock, one host read of the loss record per %d iterations" % log_every}
    if ema_ref is not None:
        rep["teacher_vs_ema_of_student_maxabs"] = float((ema_ref - step.flat_t.data).abs().max())
        rep["teacher_maxabs"] = float(step.flat_t.data.abs().max())
    return step, rep


class SyntheticKitti:
    """Held-out scans as a KITTI-format validation set for det3d.datasets.kitti.KittiDataset.evaluation: one info per scan
    (image index = position, the KITTI-typical calibration of synth.kitti_calib) whose `annos` are the scan's cars converted by
    the SAME function detections go through (lidar box -> rectified camera frame, 2-D box by projection, boxes outside the image
    dropped) -- cars with >= MIN_POINTS points as fully visible `Car`s, cars with fewer (but some) points as occluded = 3
    (`unknown`: ignored at every difficulty, so that finding one is not a false positive), cars without a point not at all."""

    def __init__(self, pool):
        from det3d.datasets.kitti.kitti import KittiDataset, convert_detection_to_kitti_annos
        cal = synth.kitti_calib()
        calib = {"R0_rect": cal["rect"], "Tr_velo_to_cam": cal["Trv2c"], "P2": cal["P2"]}
        infos = []
        for i in range(len(pool)):
            info = {"image": {"image_idx": i, "image_shape": np.array(cal["image_shape"])}, "calib": calib}
            some = pool.counts[i] > 0
            boxes = pool.cars[i][some]
            det = {str(i): dict(box3d_lidar=boxes.copy(), scores=np.ones(len(boxes), np.float32), label_preds=np.zeros(len(boxes), np.int64),
                                metadata={"token": str(i), "order": np.arange(len(boxes))})}
            # the conversion drops boxes outside the image: carry each box's visibility through it by its score
            vis = (pool.counts[i][some] >= MIN_POINTS).astype(np.float32)
            det[str(i)]["scores"] = 0.25 + 0.5 * vis
            anno = convert_detection_to_kitti_annos(det, [info], ["Car"])[0]
            anno["occluded"] = np.where(np.asarray(anno["score"], np.float32) > 0.5, 0, 3).astype(np.int64) if len(anno["name"]) else anno["occluded"]
            anno.pop("score"); anno.pop("metadata", None)
            info["annos"] = anno
            infos.append(info)
        self.infos = infos
        self.dataset = KittiDataset(kitti_infos=infos, class_names=["Car"], test_mode=True)

    def evaluate(self, detections):
        """detections: list (per scan, in pool order) of dict(box3d_lidar, scores, label_preds). Returns dict(ap3d_11, ap3d_40,
        bev_11, ...: [easy, moderate, hard] at the 0.7 overlap of the car class) + the printed official table."""
        det = {str(i): dict(box3d_lidar=d["box3d_lidar"], scores=d["scores"], label_preds=d["label_preds"], metadata={"token": str(i)})
               for i, d in enumerate(detections)}
        res, _ = self.dataset.evaluation(det)
        off = res["detail"]["eval.kitti"]["official"]["car"]
        from det3d.datasets.kitti.eval import get_official_eval_result_v2
        dt_annos = self.dataset.convert_detection_to_kitti_annos(det)
        r40 = get_official_eval_result_v2(self.dataset.ground_truth_annotations, dt_annos, ["Car"], z_axis=1, z_center=1.0)["detail"]["car"]
        return {"ap3d_11": off["3d@0.70"], "bev_11": off["bev@0.70"], "bbox_11": off["bbox@0.70"], "ap3d_40": r40["3d@0.70"], "bev_40": r40["bev@0.70"],
                "table": res["results"]["official_AP_11"],
                "gt_cars": int(sum(int((np.asarray(a["annos"]["occluded"]) == 0).sum()) for a in self.infos)),
                "detections": int(sum(len(d["scores"]) for d in detections))}
