"""sessd_hip.runner.HostFedPipeline: host point clouds in, host detections out, no host synchronisation per frame (the loop of
the reference's tools/test.py:121-146 as a pipeline). Every frame's detections must equal what the same engines return one
frame at a time, in submission order -- with pinned and pageable inputs, varying point counts, more frames than the staging ring
and the record ring hold, through captured graphs and eagerly, and after reset()."""
import numpy as np
import pytest
import torch

from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
from sessd_hip.runner import HostFedPipeline

pytestmark = pytest.mark.gpu
VG = configs.VOXEL_GENERATOR


@pytest.fixture(scope="module")
def model(dev):
    return configs.build_synthetic_detector(dev, seed=0)


@pytest.mark.parametrize("eager,copy_mode", [(False, "instream"), (True, "instream"), (False, "copystream")])
def test_pipeline_returns_every_frame_in_order(dev, model, eager, copy_mode):
    """copy_mode: the H2D of a frame on its engine's own stream (round 5 default) or on a copy stream with an event the engine waits
    for (rounds 2 - 4)"""
    engines = [InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev) for _ in range(2)]
    frames = [synth.make_frame(70 + i, 20000 - 700 * (i % 5)) for i in range(11)]
    # reference: one frame at a time on engine 0 (both engines share the configuration, so their bits agree)
    want = []
    for f in frames:
        engines[0].set_points([torch.from_numpy(f).to(dev)])
        engines[0].enqueue()
        want.append(engines[0].results()[0])
    pipe = HostFedPipeline(engines, ring=3, fetch_every=4, eager=eager, copy_mode=copy_mode)   # rings of 8 records: 11 + 9 frames wrap them
    if not eager:
        for e in engines:
            e.capture()
    n_frames = 20
    got = []
    for rnd in range(2):
        pipe.reset()
        got = []
        for i in range(n_frames):
            f = frames[i % len(frames)]
            src = torch.from_numpy(f).pin_memory() if i % 3 == 0 else f      # pinned tensors are used in place, numpy is staged
            assert pipe.submit(src) == i
            if i % 5 == 4:
                got += pipe.poll()
        got += pipe.finish()
        assert len(got) == n_frames
        for i, g in enumerate(got):
            w = want[i % len(frames)]
            for k in ("box3d_lidar", "scores", "label_preds"):
                assert np.array_equal(g[k], w[k]), (rnd, i, k)
    assert sum(len(g["scores"]) for g in got) > 100


def test_pipeline_rejects_what_it_cannot_feed(dev, model):
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 2, 20480, dev)
    with pytest.raises(AssertionError):
        HostFedPipeline([eng])           # batch-1 engines only
    eng1 = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 4096, dev)
    pipe = HostFedPipeline([eng1], fetch_every=2, eager=True)
    with pytest.raises(ValueError):
        pipe.submit(np.zeros((5000, 4), np.float32))    # more points than the engine's capacity
    eng1.attach_records(3)
    eng1.capture()
    with pytest.raises(RuntimeError):
        HostFedPipeline([eng1], fetch_every=2)          # ring of the wrong size baked into a captured graph


def test_pipeline_without_reset_after_capture_and_after_finish(dev, model):
    """Round-3 advisor finding: capture() (two warm-up enqueues + the capture pass) advances the device record cursor, and
    finish() can leave the fetch position off a multiple of fetch_every; the host ring arithmetic assumed neither. submit()
    on an idle pipeline now re-synchronises by itself: no reset() anywhere in this test."""
    engines = [InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev) for _ in range(2)]
    frames = [synth.make_frame(90 + i, 20000 - 900 * (i % 4)) for i in range(7)]
    want = []
    for f in frames:
        engines[0].set_points([torch.from_numpy(f).to(dev)])
        engines[0].enqueue()
        want.append(engines[0].results()[0])
    pipe = HostFedPipeline(engines, ring=2, fetch_every=4)
    for e in engines:
        e.capture()                      # cursor != 0 from here on
    engines[1].set_points([torch.from_numpy(frames[0]).to(dev)])
    engines[1].replay()                  # and some eager use on top
    idx = 0
    for job, n_frames in enumerate((7, 5, 9)):   # 7 and 5 frames over two engines: fetch positions 4+3, then 3+2 -- never multiples of 4
        got = []
        for i in range(n_frames):
            assert pipe.submit(frames[i % len(frames)]) == idx
            idx += 1
        got += pipe.finish()
        assert len(got) == n_frames
        for i, g in enumerate(got):
            for k in ("box3d_lidar", "scores", "label_preds"):
                assert np.array_equal(g[k], want[i % len(frames)][k]), (job, i, k)


def test_pipeline_overflow_of_an_earlier_frame_is_not_lost(dev, model):
    """Round-3 advisor finding (medium): the sparse-level overflow flag lived in the arena every frame clears, so the pipeline --
    which reads it once at the end -- lost an overflow of any frame but the last and returned truncated detections. The flag
    is sticky now and travels with every record fetch: the pipeline raises, results() raises once and re-arms."""
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev,
                          growth=(0.25, 1.0, 0.75, 0.75))   # level 1 holds a quarter of the voxels: a full scan overflows it
    big, small = synth.make_frame(3, 20000), synth.make_frame(4, 20000)[:1500].copy()
    eng.set_points([torch.from_numpy(small).to(dev)])
    eng.enqueue()
    eng.results()                        # the small frame fits
    eng.set_points([torch.from_numpy(big).to(dev)])
    eng.enqueue()
    eng.set_points([torch.from_numpy(small).to(dev)])
    eng.enqueue()                        # a later frame that fits must not erase the flag
    with pytest.raises(RuntimeError, match="overflow"):
        eng.results()
    eng.enqueue()
    eng.results()                        # re-armed: the small frame alone is fine again
    pipe = HostFedPipeline([eng], ring=2, fetch_every=2, eager=True)
    with pytest.raises(RuntimeError, match="overflow"):
        for f in (small, big, small, small, small, small):
            pipe.submit(f)
        pipe.finish()
    pipe.reset()
    for f in (small, small, small):
        pipe.submit(f)
    assert len(pipe.finish()) == 3
