"""Generates tests/golden/kitti_convert_ref.npz by running the REFERENCE's detection -> KITTI annotation conversion from source:
    det3d/datasets/kitti/kitti.py:71-139  KittiDataset.convert_detection_to_kitti_annos (called unbound on a stand-in object that
    carries `_class_names` and `_kitti_infos`), with det3d/core/bbox/box_np_ops.py box_lidar_to_camera / center_to_corner_box3d /
    project_to_image / limit_period and kitti_common.get_start_result_anno / empty_result_anno.
Stubs: dataset base class and registry (inert), skimage / tqdm (unused here), numba (identity decorators).
Run in the build container only:  python tests/golden/make_golden_kitti_convert.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
sys.path.insert(0, HERE)


def make_case():
    from sessd_hip import synth
    cal = synth.kitti_calib()
    infos, dets = [], {}
    for f in range(4):
        tok = "%06d" % (f * 7)
        infos.append(dict(image=dict(image_idx=tok, image_shape=np.array(cal["image_shape"], np.int32)),
                          calib=dict(R0_rect=cal["rect"], Tr_velo_to_cam=cal["Trv2c"], P2=cal["P2"])))
        n = [6, 0, 9, 3][f]
        b = synth.random_boxes7(n, seed=f + 1).astype(np.float32)
        if n:
            b[0, :2] = [3.0, 30.0]      # far to the side: projects outside the image (dropped)
            b[-1, 6] = 4.5              # yaw outside [-pi, pi): folded
        rng = np.random.RandomState(f)
        dets[tok] = dict(box3d_lidar=torch.from_numpy(b), scores=torch.from_numpy(rng.uniform(0.3, 1, n).astype(np.float32)),
                         label_preds=torch.zeros(n, dtype=torch.int64), metadata=dict(token=tok))
    return infos, dets


def main():
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    MG.install_stubs()
    MG.load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = MG.load_ref("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    sys.modules["det3d.core.bbox"].box_np_ops = bnp

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in kw.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        @staticmethod
        def register_module(obj):
            return obj

    stub("skimage", io=None); stub("skimage.io"); stub("tqdm", tqdm=lambda x, **k: x)
    stub("det3d.datasets"); stub("det3d.datasets.kitti")
    stub("det3d.datasets.custom", PointCloudDataset=object)
    stub("det3d.datasets.registry", DATASETS=_Reg())
    MG.load_ref("det3d/datasets/kitti/kitti_common.py", "det3d.datasets.kitti.kitti_common")
    stub("det3d.datasets.kitti.eval", get_official_eval_result=None, get_coco_eval_result=None, get_official_eval_result_v2=None)
    K = MG.load_ref("det3d/datasets/kitti/kitti.py", "ref_kitti")
    infos, dets = make_case()
    fake = types.SimpleNamespace(_class_names=["Car"], _kitti_infos=infos)
    annos = K.KittiDataset.convert_detection_to_kitti_annos(fake, dets)
    out = {}
    for i, a in enumerate(annos):
        for k in ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score"):
            v = a[k]
            out["%d_%s" % (i, k)] = np.array([str(s) for s in v]) if k == "name" else np.asarray(v, np.float64)
        print(i, a["name"].shape[0], a["metadata"])
    np.savez_compressed(os.path.join(HERE, "kitti_convert_ref.npz"), **out)


if __name__ == "__main__":
    main()
