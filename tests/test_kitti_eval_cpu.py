"""det3d.datasets.kitti.eval (mirror) vs tests/golden/kitti_eval_ref.npz = the reference's evaluation run from
source on the same 24 synthetic frames. The rotated overlaps, which the product takes from the device kernel
(sessd_rotate_iou_eval, tested against the same oracle in tests/test_rotate_iou_numba_gpu.py), are served here by the CPU oracle
oracle/rotate_iou_eval.c (bit-equal to the reference's numba device functions): everything else is the code under test."""
import os
import sys

import numpy as np

from oracle import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _oracle_rotate_iou(boxes, query_boxes, criterion=-1, device_id=0):
    if boxes.shape[0] == 0 or query_boxes.shape[0] == 0:
        return np.zeros((boxes.shape[0], query_boxes.shape[0]), np.float32)
    return capi.rotate_iou_eval(boxes.astype(np.float32), query_boxes.astype(np.float32), criterion).astype(boxes.dtype)


def test_official_result_matches_reference_run(golden_dir, monkeypatch):
    from make_golden_kitti_eval import make_annos
    import det3d.datasets.utils.eval as U
    from det3d.datasets.kitti import eval as K
    monkeypatch.setattr(U, "rotate_iou_gpu_eval", _oracle_rotate_iou)
    g = np.load(os.path.join(golden_dir, "kitti_eval_ref.npz"))
    gts, dts = make_annos()
    res = K.get_official_eval_result(gts, dts, ["Car", "Pedestrian"])
    seen = 0
    for cls, d in res["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["%s|%s" % (cls, k)], rtol=0, atol=1e-9), (cls, k, v, g["%s|%s" % (cls, k)])
            seen += 1
    assert seen == len([k for k in g.files if k.count("|") == 1]) and "car AP(Average Precision)@0.70, 0.70, 0.70:" in res["result"]
    r40 = K.get_official_eval_result_v2(gts, dts, ["Car", "Pedestrian"])
    for cls, d in r40["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["r40|%s|%s" % (cls, k)], rtol=0, atol=1e-9), (cls, k)
    coco = K.get_coco_eval_result(gts, dts, ["Car", "Pedestrian"])
    for cls, d in coco["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["coco|%s|%s" % (cls, k)], rtol=0, atol=1e-9), (cls, k)
    assert coco["result"] == str(g["coco_text"])
    min_overlaps = np.array([[[0.7, 0.5], [0.7, 0.5], [0.7, 0.5]], [[0.7, 0.5], [0.5, 0.25], [0.5, 0.25]]])
    for metric in (0, 1, 2):
        r = K.eval_class_v3(gts, dts, [0, 1], [0, 1, 2], metric, min_overlaps, compute_aos=(metric == 0))
        assert np.allclose(r["precision"], g["precision_m%d" % metric], atol=1e-12, equal_nan=True)
        assert np.allclose(r["thresholds"], g["thresholds_m%d" % metric], atol=1e-12)
        if metric == 0:
            assert np.allclose(r["orientation"], g["aos_m0"], atol=1e-12, equal_nan=True)
    assert abs(float(K.get_mAP_v2(np.ones((41,)))) - 100.0) < 1e-12 and abs(float(K.get_mAP(np.ones((41,)))) - 100.0) < 1e-12


def test_overlap_building_blocks(monkeypatch):
    import det3d.datasets.utils.eval as U
    monkeypatch.setattr(U, "rotate_iou_gpu_eval", _oracle_rotate_iou)
    a = np.array([[0, 0, 10, 10], [5, 5, 15, 20.0]])
    b = np.array([[0, 0, 10, 10], [20, 20, 30, 30.0], [8, 8, 12, 12]])
    iou = U.image_box_overlap(a, b)
    assert iou[0, 0] == 1.0 and iou[0, 1] == 0.0 and abs(iou[0, 2] - 4.0 / (100 + 16 - 4)) < 1e-12
    assert abs(U.image_box_overlap(a, b, 0)[0, 2] - 4.0 / 100) < 1e-12 and abs(U.image_box_overlap(a, b, 1)[0, 2] - 4.0 / 16) < 1e-12
    # 3-D overlap of a camera-frame box (height axis 1, location at the bottom face) with a copy shifted up by half its height
    # (identical rotated boxes are ill-conditioned in the reference's numba intersection itself: shift along x as well)
    box = np.array([[0.0, 1.5, 10.0, 4.0, 1.5, 1.6, 0.0]])
    up = box.copy(); up[0, 1] -= 0.75; up[0, 0] += 1.0
    v = U.box3d_overlap(box, up, z_axis=1, z_center=1.0)[0, 0]
    assert abs(v - 3.6 / (19.2 - 3.6)) < 1e-5   # footprint overlap 3.0 x 1.6, height overlap 0.75, volumes 9.6 each
    assert U.box3d_overlap(box, box + np.array([[0, 5.0, 0, 0, 0, 0, 0]]), z_axis=1, z_center=1.0)[0, 0] == 0.0
    assert U.get_split_parts(10, 4) == [2, 2, 2, 2, 2] and U.get_split_parts(8, 4) == [2, 2, 2, 2]


def test_detection_to_kitti_annos_matches_reference_run(golden_dir):
    """det3d.datasets.kitti.kitti.convert_detection_to_kitti_annos (mirror; pure numpy) vs the reference method run from source."""
    from make_golden_kitti_convert import make_case
    from det3d.datasets.kitti.kitti import KittiDataset
    g = np.load(os.path.join(golden_dir, "kitti_convert_ref.npz"))
    infos, dets = make_case()
    annos = KittiDataset(kitti_infos=infos, class_names=["Car"]).convert_detection_to_kitti_annos(dets)
    assert len(annos) == 4 and [a["name"].shape[0] for a in annos] == [4, 0, 7, 2]
    for i, a in enumerate(annos):
        assert list(a["name"]) == list(g["%d_name" % i]) and a["metadata"] == dict(token="%06d" % (i * 7))
        for k in ("truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score"):
            assert np.allclose(np.asarray(a[k], np.float64), g["%d_%s" % (i, k)], rtol=1e-12, atol=1e-12), (i, k)


def test_dataset_evaluation_surface(monkeypatch):
    """KittiDataset.evaluation (reference kitti.py:141-166): conversion + the three result tables; detections scored against
    annotations derived from themselves (nudged by centimetres) have precision 1 wherever recall is sampled."""
    import det3d.datasets.utils.eval as U
    monkeypatch.setattr(U, "rotate_iou_gpu_eval", _oracle_rotate_iou)
    from make_golden_kitti_convert import make_case
    from det3d.datasets.kitti import eval as K
    from det3d.datasets.kitti.kitti import KittiDataset
    infos, dets = make_case()
    for info, a in zip(infos, KittiDataset(kitti_infos=infos, class_names=["Car"]).convert_detection_to_kitti_annos(dets)):
        gt = {k: np.array(v, copy=True) for k, v in a.items() if k not in ("metadata", "score")}
        gt["bbox"] = gt["bbox"].reshape(-1, 4).copy()
        if gt["bbox"].shape[0]:
            gt["location"] = gt["location"] + np.array([0.03, 0.0, 0.02])   # identical rotated boxes are ill-conditioned
        info["annos"] = gt
    ds = KittiDataset(kitti_infos=infos, class_names=["Car"])
    results, dt_annos = ds.evaluation(dets)
    assert len(dt_annos) == len(ds) == 4
    assert set(results) == {"results", "results_2", "detail"} and set(results["detail"]["eval.kitti"]) == {"official", "coco"}
    assert "car AP(Average Precision)@0.70, 0.70, 0.70:" in results["results"]["official_AP_11"]
    assert "car AP(Average Precision)@0.70, 0.70, 0.70:" in results["results_2"]["official_AP_40"]
    # a dozen objects give a dozen recall samples of the 41 (one threshold per matched detection, eval.py get_thresholds):
    # every reached sample has precision 1, i.e. AP = reached samples / 11 -- the same on every metric
    car = results["detail"]["eval.kitti"]["official"]["car"]
    assert car["bev@0.70"] == car["3d@0.70"] == car["bbox@0.70"] and min(car["bev@0.70"]) > 0
    assert all(abs(v * 11 / 100 - round(v * 11 / 100)) < 1e-9 for v in car["bev@0.70"])
    direct = K.get_official_eval_result(ds.ground_truth_annotations, dt_annos, ["Car"])
    assert direct["detail"] == results["detail"]["eval.kitti"]["official"] and direct["result"] == results["results"]["official_AP_11"]
    assert min(results["detail"]["eval.kitti"]["coco"]["car"]["bev"]) > 0
    assert ds.evaluation(dets, get_results=False)[0] is None
