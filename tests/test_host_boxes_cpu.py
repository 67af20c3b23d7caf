"""The host functions of libsessd_hip.so for the box-level decisions of the training data path (csrc/host_boxes.hip:
sessd_box_collision_host, sessd_noise_per_box_host) against the vectorised numpy forms they replace
(det3d/core/sampler/preprocess.py: box_collision_test_numpy, the Python loop of noise_per_box) -- identical decisions on random
scenes in float32 and float64, including touching / nested / far-apart quads. The numpy forms are the ones pinned to the
reference's own code by tests/test_datapath_cpu.py."""
import numpy as np
import pytest

from det3d.core.bbox import box_np_ops
from det3d.core.sampler import preprocess as prep


def _quads(rng, n, dtype, spread):
    boxes = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(1.2, 2.2, n), rng.uniform(3.0, 5.0, n),
                      rng.uniform(-np.pi, np.pi, n)], 1).astype(dtype)
    return boxes, box_np_ops.box2d_to_corner_jit(boxes)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("seed,spread", [(0, 8.0), (1, 20.0), (2, 3.0), (3, 60.0)])
def test_collision_matrix_equals_numpy(dtype, seed, spread):
    assert prep._native() is not None, "libsessd_hip.so must load (host functions need no GPU)"
    rng = np.random.RandomState(seed)
    _, a = _quads(rng, 37, dtype, spread)
    _, b = _quads(rng, 23, dtype, spread)
    b[:5] = a[:5]                                   # identical quads
    b[5:8] = a[5:8] * dtype(0.5) + a[5:8].mean(1, keepdims=True) * dtype(0.5)   # nested
    b[8] = a[8] + np.array([a[8, :, 0].max() - a[8, :, 0].min(), 0], dtype)     # touching bounding rectangles
    for cw in (True, False):
        want = prep.box_collision_test_numpy(a, b, cw)
        got = prep.box_collision_test(a, b, cw)
        assert got.dtype == np.bool_ and np.array_equal(got, want)
    assert prep.box_collision_test(a[:0], b).shape == (0, 23)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("seed,n,spread,tries", [(0, 30, 25.0, 100), (1, 12, 6.0, 100), (2, 40, 10.0, 5), (3, 1, 5.0, 10)])
def test_noise_per_box_equals_the_python_loop(dtype, seed, n, spread, tries):
    rng = np.random.RandomState(100 + seed)
    boxes, _ = _quads(rng, n, dtype, spread)
    valid = rng.rand(n) > 0.2
    loc = rng.normal(scale=1.0, size=(n, tries, 3))
    rot = rng.uniform(-0.78, 0.78, size=(n, tries))
    got = prep.noise_per_box(boxes.copy(), valid, loc, rot)
    prep.USE_NATIVE_BOX_OPS = False
    try:
        want = prep.noise_per_box(boxes.copy(), valid, loc, rot)
    finally:
        prep.USE_NATIVE_BOX_OPS = True
    assert np.array_equal(got, want)
    assert (got[~valid] == -1).all()
    if n > 10 and spread < 10:
        assert (got[valid] != 0).any()              # crowded scene: some first candidates collide
