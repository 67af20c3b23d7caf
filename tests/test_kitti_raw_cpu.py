"""KITTI raw formats on either side of the path (label_2 / calib / velodyne files -> info entries and reduced point clouds;
detections -> result lines): the det3d mirror vs tests/golden/kitti_raw_ref.npz = the reference's kitti_common.py run from source
on the same synthetic KITTI tree (tests/golden/make_golden_kitti_raw.py)."""
import os
import pickle
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "kitti_raw_ref.npz"))


def test_info_entries_and_reduced_clouds_match_reference_run(G):
    from make_golden_kitti_raw import ANNO_KEYS, make_kitti_tree
    from det3d.datasets.kitti import kitti_common as K
    with tempfile.TemporaryDirectory() as tmp:
        make_kitti_tree(tmp)
        out = K.create_kitti_info_file(tmp, splits=dict(train=[0, 1, 2], val=[1], test=[0]))
        infos = out["train"]
        for i, info in enumerate(infos):
            for k in ANNO_KEYS:
                got, want = np.asarray(info["annos"][k]), G["%d_%s" % (i, k)]
                assert got.dtype == want.dtype and got.shape == want.shape, (i, k, got.dtype, want.dtype)
                assert np.array_equal(got, want), (i, k)
            assert set(info["calib"]) == {"P0", "P1", "P2", "P3", "R0_rect", "Tr_velo_to_cam", "Tr_imu_to_velo"}
            for k, v in info["calib"].items():
                assert v.shape == (4, 4) and np.array_equal(v, G["%d_calib_%s" % (i, k)])
            assert np.array_equal(info["image"]["image_shape"], G["%d_shape" % i]) and info["image"]["image_shape"].dtype == np.int32
            assert [info["image"]["image_path"], info["point_cloud"]["velodyne_path"]] == list(G["%d_paths" % i])
        assert sorted(out["test"][0].keys()) == list(G["test_keys"]) and "annos" not in out["test"][0]
        assert len(out["trainval"]) == 4 and out["val"][0]["image"]["image_idx"] == 1
        for name in ("train", "val", "trainval", "test"):
            assert len(pickle.load(open(os.path.join(tmp, "kitti_infos_%s.pkl" % name), "rb"))) == len(out[name])
        K.create_reduced_point_cloud(tmp)
        for i in range(3):
            red = np.fromfile(os.path.join(tmp, "training/velodyne_reduced/%06d.bin" % i), dtype=np.float32).reshape(-1, 4)
            assert np.array_equal(red, G["%d_reduced" % i]) and 0 < red.shape[0] < 4500
        assert os.path.exists(os.path.join(tmp, "testing/velodyne_reduced/000000.bin"))
        # the loaders then prefer the reduced cloud, and the dataset entry runs on these files
        from det3d.datasets.kitti.kitti import KittiDataset
        ds = KittiDataset(tmp, os.path.join(tmp, "kitti_infos_train.pkl"), test_mode=True, class_names=["Car"],
                          pipeline=[dict(type="LoadPointCloudFromFile"), dict(type="LoadPointCloudAnnotations", with_bbox=True)])
        r = ds[2]
        assert np.array_equal(r["lidar"]["points"], G["2_reduced"]) and "DontCare" not in r["lidar"]["annotations"]["names"]
        assert len(r["lidar"]["annotations"]["names"]) == 7


def test_result_lines(G):
    from det3d.datasets.kitti import kitti_common as K
    a = {k: G["2_" + k] for k in ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")}
    assert K.annos_to_kitti_label(a) == list(G["label_lines"])
    assert K.kitti_result_line(dict(name="Car", bbox=[1.5, 2, 3, 4.25])) == str(G["line_defaults"])
    with pytest.raises(ValueError):
        K.kitti_result_line(dict(name="Car", bbox=None))
    with pytest.raises(KeyError):
        K.kitti_result_line(dict(name="Car", bbox=[0, 0, 1, 1], colour="red"))
    with tempfile.TemporaryDirectory() as tmp:
        det = dict(a, metadata=dict(image_idx=12))
        K.kitti_anno_to_label_file([det], tmp)
        lines = open(os.path.join(tmp, "000012.txt")).read().split("\n")
        assert len(lines) == len(a["name"]) and lines[0].split(" ")[0] == a["name"][0] and lines[0].split(" ")[1:3] == ["-1", "-1"]
        back = K.get_label_anno(os.path.join(tmp, "000012.txt"))   # 16 columns: the score is read back
        assert np.allclose(back["score"], a["score"], atol=1e-4) and np.allclose(back["dimensions"], a["dimensions"], atol=1e-4)


def test_difficulty_levels():
    from det3d.datasets.kitti import kitti_common as K
    bbox = lambda h: [0.0, 0.0, 50.0, h]
    info = dict(annos=dict(bbox=np.array([bbox(41), bbox(41), bbox(30), bbox(26), bbox(25), bbox(41)]), dimensions=np.zeros((6, 3)),
                           occluded=np.array([0, 1, 0, 2, 0, 3]), truncated=np.array([0.1, 0.1, 0.0, 0.4, 0.0, 0.0])))
    assert K.add_difficulty_to_annos(info) == [0, 1, 1, 2, -1, -1] and info["annos"]["difficulty"].dtype == np.int32
