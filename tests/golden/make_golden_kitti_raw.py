"""Generates tests/golden/kitti_raw_ref.npz by running the REFERENCE's KITTI raw-format code from source on a synthetic KITTI tree:
    det3d/datasets/kitti/kitti_common.py  get_label_anno :824-860, add_difficulty_to_annos :733-771, get_kitti_image_info :364-451
                                          (calibration parsing :408-436), _calculate_num_points_in_gt :62-92,
                                          _create_reduced_point_cloud :154-185, kitti_result_line :661-710, annos_to_kitti_label :713-730
    det3d/core/bbox/box_np_ops.py         remove_outside_points :981-992
skimage (only used for the image shape) is stubbed by a PNG-header reader; numba kernels run as Python loops.
Run in the build container only:  python tests/golden/make_golden_kitti_raw.py"""
import os
import struct
import sys
import tempfile
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CALIB_TXT = """P0: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 0.000000000000e+00 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P1: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.875744000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P2: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 4.485728000000e+01 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.163791000000e-01 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.745884000000e-03
P3: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.395242000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.199936000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.729905000000e-03
R0_rect: 9.999239000000e-01 9.837760000000e-03 -7.445048000000e-03 -9.869795000000e-03 9.999421000000e-01 -4.278459000000e-03 7.402527000000e-03 4.351614000000e-03 9.999631000000e-01
Tr_velo_to_cam: 7.533745000000e-03 -9.999714000000e-01 -6.166020000000e-04 -4.069766000000e-03 1.480249000000e-02 7.280733000000e-04 -9.998902000000e-01 -7.631618000000e-02 9.998621000000e-01 7.523790000000e-03 1.480755000000e-02 -2.717806000000e-01
Tr_imu_to_velo: 9.999976000000e-01 7.553071000000e-04 -2.035826000000e-03 -8.086759000000e-01 -7.854027000000e-04 9.998898000000e-01 -1.482298000000e-02 3.195559000000e-01 2.024406000000e-03 1.482454000000e-02 9.998881000000e-01 -7.997231000000e-01
"""


def write_png(path, h, w):
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + b"\x00" * w for _ in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def png_hw(path):
    with open(path, "rb") as f:
        head = f.read(24)
    w, h = struct.unpack(">II", head[16:24])
    return h, w


def make_kitti_tree(root, seed=0):
    """training/{image_2,label_2,calib,velodyne} for ids 0..2 and testing/{image_2,calib,velodyne} for id 0."""
    rng = np.random.RandomState(seed)
    for split, ids in (("training", (0, 1, 2)), ("testing", (0,))):
        for d in ("image_2", "calib", "velodyne") + (("label_2",) if split == "training" else ()):
            os.makedirs(os.path.join(root, split, d), exist_ok=True)
        for i in ids:
            write_png(os.path.join(root, split, "image_2", "%06d.png" % i), 375 - i, 1242 - 2 * i)
            with open(os.path.join(root, split, "calib", "%06d.txt" % i), "w") as f:
                f.write(CALIB_TXT)
            n = 4000
            pts = np.stack([rng.uniform(-10, 70, n), rng.uniform(-40, 40, n), rng.uniform(-2.5, 1.0, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
            if split == "training":
                lines, k = [], 5 + i
                for j in range(k):
                    name = ["Car", "Pedestrian", "Van", "Cyclist"][j % 4]
                    x, y, z = rng.uniform(-12, 12), rng.uniform(1.3, 1.9), rng.uniform(8, 50)
                    h, w, l = rng.uniform(1.4, 1.7), rng.uniform(1.5, 1.8), rng.uniform(3.4, 4.4)
                    ry = rng.uniform(-3.1, 3.1)
                    top = rng.uniform(100, 250)
                    hpx = [60.0, 30.0, 20.0, 45.0][j % 4]
                    left = rng.uniform(0, 1000)
                    lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" % (
                        name, [0.0, 0.2, 0.4, 0.6][(j + i) % 4], (j + i) % 3, rng.uniform(-3, 3), left, top, left + 80, top + hpx,
                        h, w, l, x, y, z, ry))
                    # a cluster of lidar points at the object (camera x,y,z -> lidar approx x=z, y=-x, z=-y+0.8)
                    c = np.array([z + 0.27, -x, -y + 0.8 + 0.4])
                    obj = (rng.uniform(-0.4, 0.4, (60, 3)) * [1.2, 0.6, 0.5] + c)
                    pts = np.concatenate([pts, np.concatenate([obj, rng.uniform(0, 1, (60, 1))], 1).astype(np.float32)])
                lines.append("DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10")
                lines.append("DontCare -1 -1 -10 511.35 174.96 527.81 187.45 -1 -1 -1 -1000 -1000 -1000 -10")
                with open(os.path.join(root, split, "label_2", "%06d.txt" % i), "w") as f:
                    f.write("\n".join(lines) + "\n")
            pts.tofile(os.path.join(root, split, "velodyne", "%06d.bin" % i))


ANNO_KEYS = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score", "index", "group_ids",
             "difficulty", "num_points_in_gt")


def main():
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    MG.install_stubs()
    for name in ("skimage", "tqdm", "det3d.datasets", "det3d.datasets.kitti"):
        sys.modules.setdefault(name, types.ModuleType(name))
    io = types.ModuleType("skimage.io")
    io.imread = lambda p: np.zeros(png_hw(p) + (3,), np.uint8)
    sys.modules["skimage"].io = io
    sys.modules["skimage.io"] = io
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    np.bool = np.bool_   # removed alias used by add_difficulty_to_annos
    geo = MG.load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = MG.load_ref("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    sys.modules["det3d.core.bbox"].box_np_ops = bnp
    kc = MG.load_ref("det3d/datasets/kitti/kitti_common.py", "det3d.datasets.kitti.kitti_common")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        make_kitti_tree(tmp)
        infos = kc.get_kitti_image_info(tmp, training=True, label_info=True, velodyne=True, calib=True, image_ids=[0, 1, 2], relative_path=True)
        kc._calculate_num_points_in_gt(tmp, infos, True)
        for i, info in enumerate(infos):
            for k in ANNO_KEYS:
                out["%d_%s" % (i, k)] = np.asarray(info["annos"][k])
            for k, v in info["calib"].items():
                out["%d_calib_%s" % (i, k)] = v
            out["%d_shape" % i] = info["image"]["image_shape"]
            out["%d_paths" % i] = np.array([info["image"]["image_path"], info["point_cloud"]["velodyne_path"]])
        test = kc.get_kitti_image_info(tmp, training=False, label_info=False, velodyne=True, calib=True, image_ids=[0], relative_path=True)
        out["test_keys"] = np.array(sorted(test[0].keys()))
        import pickle
        with open(os.path.join(tmp, "kitti_infos_train.pkl"), "wb") as f:
            pickle.dump(infos, f)
        os.makedirs(os.path.join(tmp, "training/velodyne_reduced"))
        kc._create_reduced_point_cloud(tmp, os.path.join(tmp, "kitti_infos_train.pkl"))
        for i in range(3):
            out["%d_reduced" % i] = np.fromfile(os.path.join(tmp, "training/velodyne_reduced/%06d.bin" % i), dtype=np.float32).reshape(-1, 4)
            print("frame", i, "objects", len(infos[i]["annos"]["name"]), "difficulty", infos[i]["annos"]["difficulty"].tolist(),
                  "points in gt", infos[i]["annos"]["num_points_in_gt"].tolist(), "reduced", out["%d_reduced" % i].shape[0])
        a = infos[2]["annos"]
        out["label_lines"] = np.array(kc.annos_to_kitti_label(a))
        out["line_defaults"] = np.array(kc.kitti_result_line(dict(name="Car", bbox=[1.5, 2, 3, 4.25])))
    np.savez_compressed(os.path.join(HERE, "kitti_raw_ref.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
