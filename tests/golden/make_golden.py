"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN SOURCES.

Only works where /root/reference exists (the build container). The reference package cannot be
imported as a whole here (numba, spconv, apex, ... are absent), so individual reference files are
executed from where they lie with the missing third-party modules stubbed:
  * numba.jit / njit / cuda.jit -> identity decorators (the kernels are plain Python loops)
  * spconv.utils, det3d.ops.nms.* -> empty stubs (not called by the functions used here)
Compiled pieces: det3d/core/iou3d/src/iou3d_cpu.cpp is compiled unmodified by oracle/build.py.
Nothing from the reference is copied into this repository; only INPUT/OUTPUT vectors are stored.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))


def _identity_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def install_stubs():
    numba = types.ModuleType("numba")
    numba.jit = _identity_decorator
    numba.njit = _identity_decorator
    numba.prange = range
    numba.float32 = np.float32
    numba.int32 = np.int32
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _identity_decorator

    class _Local:
        @staticmethod
        def array(shape, dtype=None):
            return np.zeros(shape, dtype=np.float32 if dtype is None else dtype)

    cuda.local = _Local
    cuda.shared = _Local
    numba.cuda = cuda
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cuda
    for name in ("spconv", "spconv.utils", "det3d", "det3d.core", "det3d.core.bbox", "det3d.ops", "det3d.ops.nms",
                 "det3d.ops.nms.nms_cpu", "det3d.ops.nms.nms_gpu", "det3d.ops.nms.nms", "det3d.utils",
                 "det3d.utils.buildtools", "det3d.utils.buildtools.pybind11_build"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["spconv.utils"].rbbox_intersection = None
    sys.modules["spconv.utils"].rbbox_iou = None
    sys.modules["det3d.ops.nms.nms"].non_max_suppression = None
    sys.modules["det3d.utils.buildtools.pybind11_build"].load_pb11 = None
    for n in ("rotate_nms_cc", "rotate_weighted_nms_cc"):
        setattr(sys.modules["det3d.ops.nms.nms_cpu"], n, None)
    for n in ("nms_gpu", "rotate_iou_gpu", "rotate_nms_gpu"):
        setattr(sys.modules["det3d.ops.nms.nms_gpu"], n, None)


def load_ref(relpath, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def main():
    assert os.path.isdir(REF), "the reference tree is only present in the build container"
    install_stubs()
    from sessd_hip import synth

    # ---------------- voxelizer: point_cloud_ops_v2.points_to_voxel (numba kernel run as Python)
    pc = load_ref("det3d/ops/point_cloud/point_cloud_ops_v2.py", "ref_point_cloud_ops_v2")
    cases = {}
    frame = synth.make_frame(seed=3, num_points=6000)
    rng = np.random.RandomState(7)
    edge = frame[:1500].copy()
    # points exactly on voxel boundaries / range limits and a few outside
    edge[:200, 0] = np.round(edge[:200, 0] / 0.05) * 0.05
    edge[200:300, 1] = -40.0
    edge[300:400, 2] = 1.0
    edge[400:420, 0] = 70.4
    edge[420:440, 0] = np.float32(70.4) - np.float32(1e-6)
    edge[440:460, 2] = -3.0
    edge[460:470, 0] = -0.01  # just below the range (NaN is undefined behaviour in the reference kernel: not pinned)
    dup = np.repeat(frame[:300], 9, axis=0)  # > max_points per voxel
    rng.shuffle(dup)
    for name, pts, mp, mv in (("frame", frame, 5, 20000), ("cap", frame, 5, 1500), ("edge", edge, 5, 20000),
                              ("dup", dup, 5, 20000), ("mp35", frame[:3000], 35, 20000),
                              ("empty", frame[:0], 5, 20000)):
        v, c, n = pc.points_to_voxel(pts, np.array(synth.KITTI_VOXEL, np.float32),
                                     np.array(synth.KITTI_RANGE, np.float32), mp, True, mv)
        cases[name + "_pts"] = pts
        cases[name + "_cfg"] = np.array([mp, mv], np.int32)
        cases[name + "_voxels"] = v
        cases[name + "_coors"] = c
        cases[name + "_num"] = n
        print("voxel case", name, pts.shape, "->", v.shape)
    np.savez_compressed(os.path.join(HERE, "voxelize_ref.npz"), **cases)

    # ---------------- iou3d: the COMPILED reference (oracle/_ref)
    import oracle
    assert oracle.ref_lib() is not None
    b7a = synth.clustered_boxes7(48, seed=11)
    b7b = synth.clustered_boxes7(40, seed=12, clusters=4)
    b7b[:4] = b7a[:4]  # identical boxes
    b7b[4, :] = b7a[4, :]
    b7b[4, 6] += np.float32(np.pi / 2)
    a5, b5 = synth.boxes7_to_bev5(b7a), synth.boxes7_to_bev5(b7b)
    literal = np.array([[0, 0, 2, 2, 0], [1, 1, 3, 3, 0], [0, 0, 2, 2, np.pi / 4], [5, 5, 6, 6, 0.3],
                        [0, 0, 2, 2, 0], [0.5, 0, 2.5, 2, 0]], np.float32)
    # odious.py:910-917 literal boxes (x,y,z,w,l,h,r)
    lit7 = np.array([[20.8845, -16.0514, -0.5310, 1.8061, 4.6556, 1.8546, 0.2290],
                     [20.8869, -15.9686, -0.5253, 1.7909, 4.6727, 1.7605, 0.2375]], np.float32)
    # CPU wrapper convention for 3-D boxes: [x1,y1,z1,x2,y2,z2,ry]
    a7, b7 = synth.boxes7_to_bev7(b7a), synth.boxes7_to_bev7(b7b)
    np.savez_compressed(
        os.path.join(HERE, "iou3d_ref.npz"), a5=a5, b5=b5, a7=a7, b7=b7, literal=literal, lit7=lit7,
        overlap=oracle.ref_boxes_overlap_bev(a5, b5), iou_bev=oracle.ref_boxes_iou_bev(a5, b5),
        iou3d_cpu=oracle.ref_boxes_iou3d(a7, b7), lit_overlap=oracle.ref_boxes_overlap_bev(literal, literal),
        lit_iou=oracle.ref_boxes_iou_bev(literal, literal),
        lit7_iou_bev=oracle.ref_boxes_iou_bev(synth.boxes7_to_bev5(lit7), synth.boxes7_to_bev5(lit7)))
    print("iou3d golden written")

    # ---------------- numpy helpers of the predict-path NMS / anchors / frustum (box_np_ops, geometry)
    load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = load_ref("det3d/core/bbox/box_np_ops.py", "ref_box_np_ops")
    geo = sys.modules["det3d.core.bbox.geometry"]
    dets = synth.clustered_boxes7(64, seed=21)[:, [0, 1, 3, 4, 6]].astype(np.float32)
    corners = bnp.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4])
    standup = bnp.corner_to_standup_nd(corners)
    standup_iou = bnp.iou_jit(standup, standup, eps=0.0)
    _mg = np.meshgrid  # numpy>=2 returns a tuple; the reference (numpy 1.x era) assigns into the result
    np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
    anchors = bnp.create_anchors_3d_range([1, 200, 176], [0, -40.0, -1.0, 70.4, 40.0, -1.0], [1.6, 3.9, 1.56],
                                          [0, 1.57])
    np.meshgrid = _mg
    cal = synth.kitti_calib()
    frustum = bnp.get_valid_frustum(cal["rect"], cal["Trv2c"], cal["P2"], cal["image_shape"])
    pts = synth.random_boxes7(400, seed=5)[:, :3]
    pts[:50, 0] = -pts[:50, 0]
    inside = geo.points_in_convex_polygon_3d_jit(pts, frustum)
    np.savez_compressed(os.path.join(HERE, "nms_helpers_ref.npz"), dets=dets, corners=corners, standup=standup,
                        standup_iou=standup_iou, anchors_sample_idx=np.arange(0, 70400, 997),
                        anchors_sample=anchors.reshape(-1, 7)[::997], anchors_sum=anchors.reshape(-1, 7).sum(0),
                        frustum=frustum, frustum_pts=pts, frustum_inside=inside)
    print("nms helper golden written; frustum keeps", int(inside.sum()), "of", len(pts))

    # ---------------- box decode (box_torch_ops.second_box_decode)
    import torch
    bto = load_ref("det3d/core/bbox/box_torch_ops.py", "ref_box_torch_ops")
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(500, 7, generator=g) * 0.3
    anc = torch.from_numpy(anchors.reshape(-1, 7)[::140][:500].copy())
    dec = bto.second_box_decode(enc, anc)
    np.savez_compressed(os.path.join(HERE, "decode_ref.npz"), enc=enc.numpy(), anchors=anc.numpy(), dec=dec.numpy())
    print("decode golden written")

    # ---------------- numba-CUDA rotated IoU device functions run as Python (nms_gpu.py:183-419)
    ng = load_ref("det3d/ops/nms/nms_gpu.py", "ref_nms_gpu")
    q = dets[:24].copy()
    ious = np.zeros((24, 24), np.float32)
    for i in range(24):
        for j in range(24):
            ious[i, j] = ng.devRotateIoU(q[i], q[j])
    evals = {}
    for crit in (-1, 0, 1, 2):
        m = np.zeros((12, 12), np.float32)
        for i in range(12):
            for j in range(12):
                m[i, j] = ng.devRotateIoUEval(q[i], q[j], crit)
        evals["eval_%d" % crit] = m
    np.savez_compressed(os.path.join(HERE, "rotate_iou_numba_ref.npz"), boxes=q, iou=ious, **{k.replace("-", "m"): v for k, v in evals.items()})
    print("numba rotate iou golden written")


if __name__ == "__main__":
    main()
