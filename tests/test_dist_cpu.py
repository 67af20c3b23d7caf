"""N>1 path on CPU: two gloo ranks shard a frame list, "detect" on their shard, and all_gather fixed-size records.
The per-frame work is replaced by a deterministic stand-in (no GPU here); the sharding / gather code is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sessd_hip import dist as sdist


def _fake_dets(frame_idx):
    rng = np.random.RandomState(frame_idx)
    n = int(rng.randint(0, 7))
    return dict(box3d_lidar=rng.rand(n, 7).astype(np.float32), scores=rng.rand(n).astype(np.float32),
                label_preds=np.zeros((n,), np.int64))


def _worker(rank, world, port, num_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx, pad = sdist.shard_indices(num_frames, rank, world)
    local = [_fake_dets(i) for i in idx]
    allf = sdist.gather_detections(local, num_frames)
    ok = all(np.array_equal(allf[i]["box3d_lidar"], _fake_dets(i)["box3d_lidar"]) and
             np.allclose(allf[i]["scores"], _fake_dets(i)["scores"]) for i in range(num_frames))
    t = torch.tensor([float(len(idx))])
    dist.all_reduce(t)  # total frames processed incl. padding
    q.put((rank, ok, len(idx), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("num_frames", [7, 8])
def test_two_rank_shard_and_gather(num_frames):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, n_local, total in res:
        assert ok, "rank %d reassembled wrong detections" % rank
        assert n_local == (num_frames + 1) // 2
        assert total == 2 * ((num_frames + 1) // 2)


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    for n, w in ((7, 2), (3769, 8), (5, 4)):
        data = list(range(n))
        for r in range(w):
            want = list(DistributedSampler(data, num_replicas=w, rank=r, shuffle=False))
            got, _ = sdist.shard_indices(n, r, w)
            assert got == want


def _worker_records(rank, world, port, num_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx, pad = sdist.shard_indices(num_frames, rank, world)
    rec, cnt = sdist.pack_detections([_fake_dets(i) for i in idx])          # what the engine leaves on the device per frame
    all_rec, all_cnt = sdist.gather_records(rec, cnt, num_frames)           # the bench's end-of-job collective
    allf = sdist.unpack_records(all_rec, all_cnt, num_frames)
    ok = all(np.array_equal(allf[i]["box3d_lidar"], _fake_dets(i)["box3d_lidar"]) and
             np.allclose(allf[i]["scores"], _fake_dets(i)["scores"]) for i in range(num_frames))
    q.put((rank, ok, tuple(all_rec.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_frames", [5, 6])
def test_two_rank_gather_of_device_records(num_frames):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_records, args=(r, world, port, num_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (2, 3, 100, 9), (rank, ok, shape)


def test_gather_records_single_process():
    rec, cnt = sdist.pack_detections([_fake_dets(i) for i in range(4)])
    all_rec, all_cnt = sdist.gather_records(rec, cnt, 4)
    assert tuple(all_rec.shape) == (1, 4, 100, 9)
    out = sdist.unpack_records(all_rec, all_cnt, 4)
    assert all(np.array_equal(out[i]["box3d_lidar"], _fake_dets(i)["box3d_lidar"]) for i in range(4))
