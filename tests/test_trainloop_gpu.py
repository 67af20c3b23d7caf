"""sessd_hip/trainloop.py on the device: the batch a DeviceBatcher assembles inside the static example equals the host assembly of
the same scenes with the same recorded augmentation (reference data flow: preprocess.py:137-140 global flip / rotation / scaling ->
Voxelization :196-232 -> AssignTarget :236-358), and the loop with the data path overlapped on a side stream trains the same bits
as the one-stream loop."""
import numpy as np
import pytest
import torch

from det3d.core.sampler import preprocess as prep
from sessd_hip import configs, ops, trainbench, trainloop

pytestmark = pytest.mark.gpu
VG = configs.VOXEL_GENERATOR


@pytest.fixture(scope="module")
def pool():
    return trainloop.ScenePool(range(700, 708), 20000)


def test_device_batch_equals_the_host_assembly(dev, pool):
    B = 4
    data = trainloop.DeviceBatcher(pool, dev, B, iterations=3, seed=5)
    ex = data.load(2)
    par, idx = data.par_host[2], data.choice[2]
    assert torch.equal(ex["transformation_dev"].cpu(), torch.from_numpy(par))
    assert par[:, 0].min() in (0.0, 1.0) and np.abs(par[:, 3]).max() <= np.pi / 4 + 1e-6 and np.all((par[:, 4] >= 0.95) & (par[:, 4] <= 1.05))
    lo, hi = np.array(VG["range"][:2]), np.array(VG["range"][3:5])
    n_stu, n_raw, pos_total = 0, 0, 0
    for b, i in enumerate(idx):
        pts, cars = pool.frames[i], pool.visible(i)
        p2, c2 = trainbench._augment(pts, cars, bool(par[b, 0]), float(par[b, 3]), float(par[b, 4]))
        got = data._student_cloud(data.frames[i], data.par[2, b]).cpu().numpy()
        assert np.abs(got - p2).max() < 2e-4
        gb = data._student_boxes(data.boxes[i][:len(cars)], data.par[2, b]).cpu().numpy()
        assert np.abs(gb - c2).max() < 2e-4
        # targets of the host-augmented boxes (range-filtered as Voxelization does) against what the static example holds
        for boxes, L in ((c2, "labels"), (cars, "labels_raw")):
            # Voxelization's filter: any BEV corner inside the range (the mirror of core/sampler/preprocess.py:138-148, itself held to
            # the reference's golden mask in tests/test_datapath_cpu.py) -- not the centre test
            keep = prep.filter_gt_box_outside_range(boxes.astype(np.float32), np.array([lo[0], lo[1], hi[0], hi[1]], np.float32))
            tg = ops.assign_targets(data.anchors, torch.from_numpy(boxes[keep]).to(dev), None, 0.6, 0.45)
            want, have = tg["labels"].cpu().numpy(), ex[L][0][b].cpu().numpy()
            # the device moved the boxes in float32, the host in float64: an anchor exactly at a matching threshold may differ
            assert int((want != have).sum()) <= 2, (b, L, int((want != have).sum()))
            pos_total += int((have > 0).sum())
        n_stu += len(np.unique(np.floor((p2[:, :3] - np.array(VG["range"][:3])) / np.array(VG["voxel_size"])).astype(np.int64)[
            np.all((p2[:, :3] >= np.array(VG["range"][:3])) & (p2[:, :3] < np.array(VG["range"][3:])), 1)], axis=0))
        n_raw += len(np.unique(np.floor((pts[:, :3] - np.array(VG["range"][:3])) / np.array(VG["voxel_size"])).astype(np.int64)[
            np.all((pts[:, :3] >= np.array(VG["range"][:3])) & (pts[:, :3] < np.array(VG["range"][3:])), 1)], axis=0))
    assert pos_total > 50
    # voxel counts: the raw cloud exactly (no arithmetic in front of the voxelizer), the student's within float32 boundary effects
    assert int(ex["num_voxels_dev_raw"].item()) == min(n_raw, B * 16000) or abs(int(ex["num_voxels_dev_raw"].item()) - n_raw) <= 4
    assert abs(int(ex["num_voxels_dev"].item()) - n_stu) <= 40
    n = int(ex["num_voxels_dev"].item())
    co = ex["coordinates"].cpu().numpy()
    assert np.all(co[:n, 0] >= 0) and np.all(co[:n, 0] < B) and np.all(co[n:] == -1) and np.all(ex["num_points"].cpu().numpy()[n:] == 1)
    assert float(ex["voxels"][n:].abs().max()) == 0.0


def test_range_filter_is_the_reference_corner_rule(dev, pool):
    """boxes straddling the range edge (centre outside, a corner inside and the reverse) through DeviceBatcher._in_range against
    det3d.core.sampler.preprocess.filter_gt_box_outside_range (reference: core/sampler/preprocess.py:138-148)"""
    data = trainloop.DeviceBatcher(pool, dev, 2, iterations=1, seed=0)
    rng = np.random.RandomState(3)
    n = 400
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = rng.choice([0.0, 70.4], n) + rng.uniform(-3, 3, n)       # around the x edges
    b[:, 1] = rng.uniform(-44, 44, n)
    b[: n // 2, 0] = rng.uniform(-3, 73, n // 2)                          # ... and around the y edges
    b[: n // 2, 1] = rng.choice([-40.0, 40.0], n // 2) + rng.uniform(-3, 3, n // 2)
    b[:, 2] = -1.0
    b[:, 3:6] = np.array([1.6, 3.9, 1.56], np.float32) * rng.uniform(0.8, 1.2, (n, 1)).astype(np.float32)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    want = prep.filter_gt_box_outside_range(b, np.array([0, -40.0, 70.4, 40.0], np.float32))
    centre = prep.filter_gt_box_outside_range_by_center(b, np.array([0, -40.0, 70.4, 40.0], np.float32))
    out = data._in_range(torch.from_numpy(b).to(dev)).cpu().numpy()
    got = out[:, 0] > trainloop.FAR / 2
    # float32 sin / cos on the device against numpy's: a corner within 1e-4 m of the edge may fall either way
    c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
    margin = np.full(n, 1e9)
    for fx, fy in ((-.5, -.5), (-.5, .5), (.5, .5), (.5, -.5)):
        x = fx * b[:, 3] * c + fy * b[:, 4] * s + b[:, 0]
        y = -fx * b[:, 3] * s + fy * b[:, 4] * c + b[:, 1]
        margin = np.minimum(margin, np.minimum(np.minimum(np.abs(x), np.abs(x - 70.4)), np.minimum(np.abs(y + 40), np.abs(y - 40))))
    clear = margin > 1e-4
    assert np.array_equal(got[clear], want[clear])
    assert int((want != centre).sum()) > 20          # the sample does separate the two rules
    assert np.array_equal(out[got], b[got])          # kept boxes pass through unchanged


def test_overlapped_data_path_trains_the_same_bits(dev, pool):
    res = []
    for overlap in (True, False):
        model = configs.build_synthetic_detector(dev, seed=0)
        step, rep = trainloop.fit(model, pool, iterations=10, batch=2, seed=3, log_every=5, overlap=overlap)
        assert rep["data_path_overlapped"] is overlap and rep["overflow_flags"] == 0
        res.append((step.flat_s.data.clone(), step.flat_t.data.clone(), [r["total"] for r in rep["log"]]))
        step.graph = None
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
