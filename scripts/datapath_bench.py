"""Times the training data path upstream of the voxelizer per sample, HOST stage against DEVICE stage:
det3d.datasets.pipelines.Preprocess (GT-AUG paste + removal of covered points, per-object noise, the teacher's points_raw
snapshot, global flip / rotation / scaling, shuffle) -> Voxelization (student + teacher view) -> AssignTarget, on synthetic
KITTI-sized frames (about 20 k points, 15 labelled boxes + 15 pasted ones). The host stage is what a DataLoader worker of the
reference does per sample (vectorised numpy instead of numba, one core); the device stage keeps the cloud in HBM from the point
file on (random draws and box-level decisions stay on the host, in the host stage's order: the same seed gives the same sample).
Prints one JSON line: ms per sample of each stage and mode, and how often the device mode reads a count back."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

from make_golden_datapath import SAMPLER_CFG, make_database, make_scene, train_cfg
from det3d.builder import build_dbsampler
from det3d.datasets.pipelines import Preprocess, Voxelization

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
VCFG = dict(range=[0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1], max_points_in_voxel=5, max_voxel_num=20000)
out = {"what": "training data path per sample: Preprocess (GT-AUG, object noise, global transform, shuffle) + Voxelization x 2",
       "samples": N}
with tempfile.TemporaryDirectory() as tmp:
    db = make_database(tmp)
    scenes = [make_scene(100 + k, n_gt=15, n_bg=15000, per_box=330) for k in range(N + 3)]
    out["points_per_frame"] = int(np.mean([s[0].shape[0] for s in scenes]))
    for mode in ("host", "device"):
        cfg = train_cfg()
        cfg["db_sampler"] = dict(SAMPLER_CFG)
        np.random.seed(7)
        pre = Preprocess(cfg=cfg, db_sampler=build_dbsampler(cfg["db_sampler"], db_infos=db))
        vox = Voxelization(cfg=VCFG)
        t_pre = t_vox = 0.0
        syncs = 0
        item = torch.Tensor.item
        if mode == "device":  # count the host reads of device scalars inside the stage
            def counting_item(self):
                global syncs
                syncs += 1
                return item(self)
            torch.Tensor.item = counting_item
        try:
            for k, (p, b, n) in enumerate(scenes):
                pts = torch.from_numpy(p).to(dev) if mode == "device" else p
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res, _ = pre(dict(labeled=True, metadata=dict(image_prefix=tmp, num_point_features=4),
                                  lidar=dict(points=pts, annotations=dict(boxes=b.copy(), names=n.copy()))), None)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                res, _ = vox(res, None)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if k >= 3:  # three warm-up samples
                    t_pre += t1 - t0
                    t_vox += t2 - t1
                elif k == 2:
                    syncs = 0
        finally:
            torch.Tensor.item = item
        out[mode] = {"preprocess_ms": t_pre / N * 1e3, "voxelization_x2_ms": t_vox / N * 1e3, "total_ms": (t_pre + t_vox) / N * 1e3}
        if mode == "device":
            out[mode]["host_reads_of_device_scalars_per_sample"] = syncs / float(N)
out["device_speedup_total"] = out["host"]["total_ms"] / out["device"]["total_ms"]
print(json.dumps(out))
