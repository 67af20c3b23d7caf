"""Generates tests/golden/kitti_eval_ref.npz by running the REFERENCE's KITTI evaluation from source on CPU:
    det3d/datasets/kitti/eval.py        get_official_eval_result :467-569, do_eval_v3 :395-421, eval_class_v3 :174-319,
                                        fused_compute_statistics :121-171, clean_data :40-108, get_thresholds :18-37, get_mAP :330-333
    det3d/datasets/utils/eval.py        calculate_iou_partly :61-140, prepare_data :18-58, compute_statistics_jit :144-278,
                                        image_box_overlap :282-312, box3d_overlap(_kernel) :324-367
numba is stubbed (kernels run as Python loops); `rotate_iou_gpu_eval` (numba-CUDA, absent) is served by the reference's own
device functions of det3d/ops/nms/nms_gpu.py executed as Python, pair by pair (the same substitution as make_golden.py).
The annotations are synthetic: 24 frames of KITTI-camera-format boxes (cars, vans, pedestrians, DontCare regions, graded
occlusion / truncation / 2-D heights) and detections = perturbed ground truth + false positives.
Run in the build container only:  python tests/golden/make_golden_kitti_eval.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
sys.path.insert(0, HERE)


def make_annos(seed=0, frames=24):
    rng = np.random.RandomState(seed)
    gts, dts = [], []
    for f in range(frames):
        n = rng.randint(5, 11)
        names = rng.choice(["Car", "Car", "Car", "Van", "Pedestrian", "DontCare"], n)
        loc = np.stack([rng.uniform(-20, 20, n), rng.uniform(1.2, 2.0, n), rng.uniform(5, 60, n)], 1)  # camera x, y, z
        dims = np.stack([rng.uniform(3.2, 4.6, n), rng.uniform(1.4, 1.8, n), rng.uniform(1.5, 1.9, n)], 1)  # l, h, w
        rot = rng.uniform(-np.pi, np.pi, n)
        h2d = rng.choice([22.0, 30.0, 45.0, 80.0, 150.0], n, p=[0.1, 0.15, 0.25, 0.25, 0.25])
        x1 = rng.uniform(0, 1000, n); y1 = rng.uniform(100, 200, n)
        bbox = np.stack([x1, y1, x1 + h2d * 1.6, y1 + h2d], 1)
        gt = dict(name=names, truncated=rng.choice([0.0, 0.0, 0.1, 0.25, 0.4, 0.8], n), occluded=rng.choice([0, 0, 0, 1, 2, 3], n),
                  alpha=rng.uniform(-np.pi, np.pi, n), bbox=bbox, dimensions=dims, location=loc, rotation_y=rot)
        gts.append(gt)
        keep = rng.rand(n) < 0.85
        keep &= names != "DontCare"
        k = int(keep.sum())
        m = rng.randint(0, 3)
        dn = np.concatenate([names[keep], rng.choice(["Car", "Pedestrian"], m)])
        dloc = np.concatenate([loc[keep] + rng.normal(0, 0.15, (k, 3)), np.stack([rng.uniform(-20, 20, m), rng.uniform(1.2, 2, m), rng.uniform(5, 60, m)], 1)])
        ddim = np.concatenate([dims[keep] * rng.uniform(0.93, 1.07, (k, 3)), np.stack([rng.uniform(3.2, 4.6, m), rng.uniform(1.4, 1.8, m), rng.uniform(1.5, 1.9, m)], 1)])
        drot = np.concatenate([rot[keep] + rng.normal(0, 0.06, k), rng.uniform(-np.pi, np.pi, m)])
        fx1 = rng.uniform(0, 1000, m); fy1 = rng.uniform(100, 200, m); fh = rng.choice([30.0, 60.0], m)
        dbox = np.concatenate([bbox[keep] + rng.normal(0, 2.0, (k, 4)), np.stack([fx1, fy1, fx1 + fh * 1.6, fy1 + fh], 1)])
        dts.append(dict(name=dn, truncated=np.zeros(k + m), occluded=np.zeros(k + m, np.int64),
                        alpha=np.concatenate([gt["alpha"][keep] + rng.normal(0, 0.1, k), rng.uniform(-np.pi, np.pi, m)]), bbox=dbox,
                        dimensions=ddim, location=dloc, rotation_y=drot, score=np.concatenate([rng.uniform(0.4, 1.0, k), rng.uniform(0.05, 0.6, m)])))
    return gts, dts


def main():
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    MG.install_stubs()
    import types
    ng = MG.load_ref("det3d/ops/nms/nms_gpu.py", "ref_nms_gpu")

    def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
        b, q = boxes.astype(np.float32), query_boxes.astype(np.float32)
        out = np.zeros((b.shape[0], q.shape[0]), np.float32)
        for i in range(b.shape[0]):
            for j in range(q.shape[0]):
                out[i, j] = ng.devRotateIoUEval(b[i], q[j], criterion)
        return out.astype(boxes.dtype)

    sys.modules["det3d.ops.nms.nms_gpu"].rotate_iou_gpu_eval = rotate_iou_gpu_eval
    sys.modules["det3d.ops.nms.nms_gpu"].inter = ng.inter
    MG.load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = MG.load_ref("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    sys.modules["det3d.core.bbox"].box_np_ops = bnp
    for name in ("det3d.datasets", "det3d.datasets.utils", "det3d.datasets.kitti"):
        m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m
    MG.load_ref("det3d/datasets/utils/eval.py", "det3d.datasets.utils.eval")
    ev = MG.load_ref("det3d/datasets/kitti/eval.py", "det3d.datasets.kitti.eval")
    gts, dts = make_annos()
    res = ev.get_official_eval_result(gts, dts, ["Car", "Pedestrian"])
    out = {}
    for cls, d in res["detail"].items():
        for k, v in d.items():
            out["%s|%s" % (cls, k)] = np.array(v, np.float64)
    r40 = ev.get_official_eval_result_v2(gts, dts, ["Car", "Pedestrian"])
    for cls, d in r40["detail"].items():
        for k, v in d.items():
            out["r40|%s|%s" % (cls, k)] = np.array(v, np.float64)
    coco = ev.get_coco_eval_result(gts, dts, ["Car", "Pedestrian"])
    for cls, d in coco["detail"].items():
        for k, v in d.items():
            out["coco|%s|%s" % (cls, k)] = np.array(v, np.float64)
    out["coco_text"] = np.array(coco["result"])
    # the raw precision / threshold arrays of one metric for a finer comparison
    min_overlaps = np.array([[[0.7, 0.5], [0.7, 0.5], [0.7, 0.5]], [[0.7, 0.5], [0.5, 0.25], [0.5, 0.25]]])
    for metric in (0, 1, 2):
        r = ev.eval_class_v3(gts, dts, [0, 1], [0, 1, 2], metric, min_overlaps, compute_aos=(metric == 0))
        out["precision_m%d" % metric] = r["precision"]
        out["thresholds_m%d" % metric] = r["thresholds"]
        if metric == 0:
            out["aos_m0"] = r["orientation"]
    np.savez_compressed(os.path.join(HERE, "kitti_eval_ref.npz"), **out)
    print(res["result"])


if __name__ == "__main__":
    main()
