"""mirrors the parts of det3d/datasets/kitti/kitti_common.py that define the data formats on either side of the path:
the KITTI raw files (label_2/*.txt, calib/*.txt, velodyne/*.bin) -> `kitti_infos_*.pkl` entries and the image-frustum-reduced
point clouds the loaders read, the annotation filters of the pipeline, and detections -> KITTI result lines.
Image sizes are read from the PNG header (the reference decodes the whole image with skimage just for its shape)."""
import pathlib
import pickle
import struct

import numpy as np

from det3d.core.bbox import box_np_ops


# ---- annotation filters used by the pipeline (:506-511, :550-553) -------------------------------------------------------------
def remove_dontcare(image_anno):
    """annotation dict without the rows named "DontCare"."""
    keep = [i for i, x in enumerate(image_anno["name"]) if x != "DontCare"]
    return {k: v[keep] for k, v in image_anno.items()}


def drop_arrays_by_name(gt_names, used_classes):
    """indices of the names NOT in used_classes."""
    return np.array([i for i, x in enumerate(gt_names) if x not in used_classes], dtype=np.int64)


def keep_arrays_by_name(gt_names, used_classes):
    return np.array([i for i, x in enumerate(gt_names) if x in used_classes], dtype=np.int64)


# ---- raw KITTI files -> info entries (:278-452, :733-771, :824-860) ---------------------------------------------------------
def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_kitti_info_path(idx, prefix, info_type="image_2", file_tail=".png", training=True, relative_path=True, exist_check=True):
    rel = pathlib.Path("training" if training else "testing") / info_type / (get_image_index_str(idx) + file_tail)
    if exist_check and not (pathlib.Path(prefix) / rel).exists():
        raise ValueError("file not exist: {}".format(rel))
    return str(rel) if relative_path else str(pathlib.Path(prefix) / rel)


def get_image_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, "image_2", ".png", training, relative_path, exist_check)


def get_label_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, "label_2", ".txt", training, relative_path, exist_check)


def get_velodyne_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, "velodyne", ".bin", training, relative_path, exist_check)


def get_calib_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, "calib", ".txt", training, relative_path, exist_check)


def png_shape(path):
    """(height, width) from the IHDR chunk of a PNG file."""
    with open(path, "rb") as f:
        head = f.read(24)
    if head[:8] != b"\x89PNG\r\n\x1a\n" or head[12:16] != b"IHDR":
        raise ValueError("not a PNG file: %s" % path)
    w, h = struct.unpack(">II", head[16:24])
    return np.array([h, w], dtype=np.int32)


def get_label_anno(label_path):
    """one label_2 file -> dict of per-object arrays; sizes reordered from the file's h,w,l to l,h,w; `index` numbers the real
    objects and is -1 for DontCare rows (which the format lists last); `score` is 0 unless a 16th column is present."""
    with open(label_path, "r") as f:
        rows = [line.strip().split(" ") for line in f.readlines()]
    col = lambda a, b: np.array([[float(v) for v in r[a:b]] for r in rows])
    n, real = len(rows), len([r for r in rows if r[0] != "DontCare"])
    anno = {"name": np.array([r[0] for r in rows]), "truncated": np.array([float(r[1]) for r in rows]),
            "occluded": np.array([int(r[2]) for r in rows]), "alpha": np.array([float(r[3]) for r in rows]),
            "bbox": col(4, 8).reshape(-1, 4), "dimensions": col(8, 11).reshape(-1, 3)[:, [2, 0, 1]],
            "location": col(11, 14).reshape(-1, 3), "rotation_y": np.array([float(r[14]) for r in rows]).reshape(-1)}
    anno["score"] = np.array([float(r[15]) for r in rows]) if n and len(rows[0]) == 16 else np.zeros((anno["bbox"].shape[0],))
    anno["index"] = np.array(list(range(real)) + [-1] * (n - real), dtype=np.int32)
    anno["group_ids"] = np.arange(n, dtype=np.int32)
    return anno


def add_difficulty_to_annos(info):
    """annos["difficulty"]: 0 easy / 1 moderate / 2 hard / -1 none, from 2-D box height (> 40 / 25 / 25 px), occlusion level
    (<= 0 / 1 / 2) and truncation (<= 0.15 / 0.3 / 0.5): the first level whose three limits hold."""
    a = info["annos"]
    height = a["bbox"][:, 3] - a["bbox"][:, 1]
    ok = [~((a["occluded"] > o) | (height <= h) | (a["truncated"] > t)) for h, o, t in ((40, 0, 0.15), (25, 1, 0.3), (25, 2, 0.5))]
    easy, moderate, hard = ok[0], np.logical_xor(ok[0], ok[1]), np.logical_xor(ok[2], ok[1])
    diff = np.where(easy, 0, np.where(moderate, 1, np.where(hard, 2, -1))).astype(np.int32)
    a["difficulty"] = diff
    return diff.tolist()


def read_calib(calib_path, extend_matrix=True):
    """calib/*.txt -> dict(P0..P3, R0_rect, Tr_velo_to_cam, Tr_imu_to_velo), 4x4 when extend_matrix (:408-436)."""
    with open(calib_path, "r") as f:
        lines = f.readlines()
    mat = lambda i, n, shape: np.array([float(v) for v in lines[i].split(" ")[1:n + 1]]).reshape(shape)
    ext = lambda m: np.concatenate([m, np.array([[0.0, 0.0, 0.0, 1.0]])], axis=0) if extend_matrix else m
    out = {"P%d" % i: ext(mat(i, 12, [3, 4])) for i in range(4)}
    r0 = mat(4, 9, [3, 3])
    if extend_matrix:
        r4 = np.zeros([4, 4], dtype=r0.dtype)
        r4[3, 3], r4[:3, :3] = 1.0, r0
        r0 = r4
    out["R0_rect"] = r0
    out["Tr_velo_to_cam"], out["Tr_imu_to_velo"] = ext(mat(5, 12, [3, 4])), ext(mat(6, 12, [3, 4]))
    return out


def get_kitti_image_info(path, training=True, label_info=True, velodyne=False, calib=False, image_ids=7481, extend_matrix=True,
                         num_worker=8, relative_path=True, with_imageshape=True):
    """one info dict per image id: image (idx, path, shape), point_cloud (num_features, velodyne_path), calib, annos (+difficulty)."""
    root = pathlib.Path(path)
    ids = image_ids if isinstance(image_ids, list) else list(range(image_ids))
    infos = []
    for idx in ids:
        info = {"image": {"image_idx": idx, "image_path": get_image_path(idx, path, training, relative_path)},
                "point_cloud": {"num_features": 4}}
        if velodyne:
            info["point_cloud"]["velodyne_path"] = get_velodyne_path(idx, path, training, relative_path)
        if with_imageshape:
            p = info["image"]["image_path"]
            info["image"]["image_shape"] = png_shape(str(root / p) if relative_path else p)
        if calib:
            info["calib"] = read_calib(get_calib_path(idx, path, training, relative_path=False), extend_matrix)
        if label_info:
            p = get_label_path(idx, path, training, relative_path)
            info["annos"] = get_label_anno(str(root / p) if relative_path else p)
            add_difficulty_to_annos(info)
        infos.append(info)
    return infos


def _calculate_num_points_in_gt(data_path, infos, relative_path, remove_outside=True, num_features=4):
    """annos["num_points_in_gt"]: lidar points (inside the image frustum) in every real object's box, -1 for DontCare (:62-92)."""
    for info in infos:
        c = info["calib"]
        v = info["point_cloud"]["velodyne_path"]
        pts = np.fromfile(str(pathlib.Path(data_path) / v) if relative_path else v, dtype=np.float32, count=-1).reshape([-1, num_features])
        if remove_outside:
            pts = box_np_ops.remove_outside_points(pts, c["R0_rect"], c["Tr_velo_to_cam"], c["P2"], info["image"]["image_shape"])
        a = info["annos"]
        real = len([n for n in a["name"] if n != "DontCare"])
        cam = np.concatenate([a["location"][:real], a["dimensions"][:real], a["rotation_y"][:real][..., np.newaxis]], axis=1)
        lidar = box_np_ops.box_camera_to_lidar(cam, c["R0_rect"], c["Tr_velo_to_cam"])
        counts = box_np_ops.points_in_rbbox(pts[:, :3], lidar).sum(0)
        a["num_points_in_gt"] = np.concatenate([counts, -np.ones([len(a["dimensions"]) - real])]).astype(np.int32)


def create_kitti_info_file(data_path, save_path=None, relative_path=True, splits=None):
    """kitti_infos_{train,val,trainval,test}.pkl from the raw files (:95-151). `splits` = dict(train=[ids], val=[ids], test=[ids])
    replaces the reference's ImageSets/*.txt lists when given."""
    if splits is None:
        sets = pathlib.Path(__file__).resolve().parent.parent / "ImageSets"
        splits = {k: [int(l) for l in open(str(sets / (k + ".txt"))).readlines()] for k in ("train", "val", "test")}
    save = pathlib.Path(data_path if save_path is None else save_path)
    out = {}
    for name in ("train", "val"):
        out[name] = get_kitti_image_info(data_path, training=True, label_info=True, velodyne=True, calib=True,
                                         image_ids=list(splits[name]), relative_path=relative_path)
        _calculate_num_points_in_gt(data_path, out[name], relative_path)
    out["trainval"] = out["train"] + out["val"]
    out["test"] = get_kitti_image_info(data_path, training=False, label_info=False, velodyne=True, calib=True,
                                       image_ids=list(splits["test"]), relative_path=relative_path)
    for name, infos in out.items():
        with open(save / ("kitti_infos_%s.pkl" % name), "wb") as f:
            pickle.dump(infos, f)
    return out


def _create_reduced_point_cloud(data_path, info_path, save_path=None, back=False):
    """velodyne/*.bin -> velodyne_reduced/*.bin: only the points the camera sees (:154-185)."""
    with open(info_path, "rb") as f:
        infos = pickle.load(f)
    for info in infos:
        c = info["calib"]
        v = pathlib.Path(data_path) / info["point_cloud"]["velodyne_path"]
        pts = np.fromfile(str(v), dtype=np.float32, count=-1).reshape([-1, 4])
        if back:
            pts[:, 0] = -pts[:, 0]
        pts = box_np_ops.remove_outside_points(pts, c["R0_rect"], c["Tr_velo_to_cam"], c["P2"], info["image"]["image_shape"])
        if save_path is None:
            dst = v.parent.parent / (v.parent.stem + "_reduced")
            dst.mkdir(parents=True, exist_ok=True)
            dst = dst / v.name
        else:
            dst = pathlib.Path(save_path) / v.name
        pts.tofile(str(dst) + ("_back" if back else ""))


def create_reduced_point_cloud(data_path, train_info_path=None, val_info_path=None, test_info_path=None, save_path=None, with_back=False):
    root = pathlib.Path(data_path)
    paths = [train_info_path or root / "kitti_infos_train.pkl", val_info_path or root / "kitti_infos_val.pkl",
             test_info_path or root / "kitti_infos_test.pkl"]
    for back in ((False, True) if with_back else (False,)):
        for p in paths:
            _create_reduced_point_cloud(data_path, p, save_path, back=back)


# ---- detections -> KITTI result files (:33-53, :661-730) -------------------------------------------------------------------
_FIELDS = (("name", None), ("truncated", -1), ("occluded", -1), ("alpha", -10), ("bbox", None), ("dimensions", [-1, -1, -1]),
           ("location", [-1000, -1000, -1000]), ("rotation_y", -10), ("score", 0.0))


def kitti_result_line(result_dict, precision=4):
    """one object as a KITTI label / result line; missing optional fields take the format's "unknown" values; sizes are
    written h, w, l from the dict's l, h, w."""
    fmt = "{" + ":.{}f".format(precision) + "}"
    defaults = dict(_FIELDS)
    for key, val in result_dict.items():
        if key not in defaults:
            raise KeyError(key)
        if defaults[key] is None and val is None:
            raise ValueError("you must specify a value for {}".format(key))
    parts = []
    for key, default in _FIELDS:
        val = result_dict.get(key)
        if key == "name":
            parts.append(val)
        elif key == "occluded":
            parts.append(str(default) if val is None else "{}".format(val))
        elif key in ("truncated", "alpha", "rotation_y", "score"):
            parts.append(str(default) if val is None else fmt.format(val))
        elif val is None:
            parts += [str(v) for v in default]
        else:
            parts += [fmt.format(v) for v in ([val[1], val[2], val[0]] if key == "dimensions" else val)]
    return " ".join(parts)


def annos_to_kitti_label(annos):
    keys = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")
    return [kitti_result_line({k: annos[k][i] for k in keys}) for i in range(len(annos["name"]))]


def kitti_anno_to_label_file(annos, folder):
    """one <image_idx>.txt per detection annotation dict (what the KITTI server / devkit reads)."""
    folder = pathlib.Path(folder)
    for anno in annos:
        keys = ("name", "alpha", "bbox", "location", "dimensions", "rotation_y", "score")
        lines = [kitti_result_line({k: anno[k][j] for k in keys}) for j in range(anno["bbox"].shape[0])]
        with open(folder / (get_image_index_str(anno["metadata"]["image_idx"]) + ".txt"), "w") as f:
            f.write("\n".join(lines))
