"""TEST INFRASTRUCTURE (run as a subprocess by tests/test_rccl_gpu.py; its EXIT CODE is part of the test).

Round-5 review, "What's weak 8": no RCCL call of this project had ever executed, and at --gpus N bench.py issues its collectives
from CU-masked streams (hipExtStreamCreateWithCUMask) and then tears down a process group with those streams still alive. A test
box has ONE GPU, so this runs a process group of one rank over the `nccl` backend (= RCCL) with SESSD_FORCE_COLLECTIVES=1, which
switches the world-1 short-circuits of the package off (sessd_hip.dist.collectives_enabled), and executes every collective the
8-GPU job would:
  * sessd_hip.dist.gather_records (all_gather_into_tensor x 2) from each of four engines' CU-masked streams, after real frames;
  * sessd_hip.train.allreduce_flat on a flat gradient buffer, on a masked stream and on the default stream;
  * the SyncBN statistics all-reduces of one eager SE-SSD training iteration (TrainStep with sync_bn=True: ~110 collectives);
  * dist.barrier();
then the ORDERED teardown bench.py uses -- graphs and engines, allocator cache, destroy_process_group, ops.close_masked_streams --
and a clean interpreter exit. Prints one JSON line of checks. Reference: tools/dist_test.py:150-186, det3d/core/utils/dist_utils.py:8-57,
det3d/torchie/apis/train_sessd.py:286-294."""
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["SESSD_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")

import numpy as np
import torch
import torch.distributed as dist


def main():
    from sessd_hip import configs, ops, runner, synth, trainbench
    from sessd_hip import dist as sdist
    from sessd_hip import train as strain
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    assert sdist.collectives_enabled()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    VG = configs.VOXEL_GENERATOR
    model = configs.build_synthetic_detector(dev, seed=0)
    engines, streams = runner.engines_on_cu_sets(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, n_engines=4, sets=2,
                                                 device=dev, capture=True, records=8)
    frames = [torch.from_numpy(synth.make_frame(i, 20000)).to(dev) for i in range(4)]
    for rep in range(2):
        for i, (e, st) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(st):
                e.set_points([frames[(i + rep) % 4]])
                e.replay()
    ok_gather = True
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            rec, cnt = sdist.gather_records(e.records, e.record_counts, 8)     # two collectives ON the masked stream
        st.synchronize()
        # (the ring also holds the records of the capture's warm-up frames, which ran on an empty cloud: the two real frames are in it)
        ok_gather &= bool(rec.shape[0] == 1 and torch.equal(rec[0], e.records[:8]) and torch.equal(cnt[0], e.record_counts[:8]))
        ok_gather &= int(cnt.sum().item()) > 0 and int(e.record_cursor.item()) >= 2
    out["gather_records_from_masked_streams"] = ok_gather
    g = torch.arange(1 << 20, dtype=torch.float32, device=dev) * 1e-3
    ref = g.clone()
    with torch.cuda.stream(streams[1]):
        strain.allreduce_flat(g)
    streams[1].synchronize()
    strain.allreduce_flat(g)
    torch.cuda.synchronize()
    out["allreduce_flat"] = bool(torch.equal(g, ref))     # one rank: divided by 1, summed over 1
    dist.barrier()
    # ---- one eager training iteration with SyncBN statistics all-reduced (the configuration TrainStep.capture refuses)
    train_model = configs.build_synthetic_detector(dev, seed=0)
    step = strain.TrainStep(train_model, None, total_steps=100)
    step.sync_bn = True
    ex, cap = trainbench.labelled_batch(dev, batch=2)
    n_before = _count_allreduce()
    loss, _, _ = step(cap, 1.0, device_schedule=True)
    torch.cuda.synchronize()
    out["syncbn_iteration_loss_finite"] = bool(torch.isfinite(loss).item())
    out["syncbn_collectives"] = _count_allreduce() - n_before
    step.check_overflow()
    dist.barrier()
    torch.cuda.synchronize()
    # ---- ordered teardown
    step = cap = ex = train_model = None
    for e in engines:
        e.graph = None
    del engines[:]
    e = st = rec = cnt = None
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    dist.destroy_process_group()
    del streams[:]
    out["masked_streams_closed"] = ops.close_masked_streams()
    out["ok"] = bool(out["gather_records_from_masked_streams"] and out["allreduce_flat"] and out["syncbn_iteration_loss_finite"]
                     and out["syncbn_collectives"] > 50 and out["masked_streams_closed"] == 4 and out["backend"] == "nccl")
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


_N = [0]


def _count_allreduce():
    return _N[0]


def _patch_counter():
    real = dist.all_reduce

    def counted(*a, **k):
        _N[0] += 1
        return real(*a, **k)
    dist.all_reduce = counted


if __name__ == "__main__":
    _patch_counter()
    sys.exit(main())
