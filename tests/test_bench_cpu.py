"""bench.py's rank function on CPU: two gloo ranks, a stub engine in place of the HIP engine (no GPU here). What is tested is
the benchmark's own control flow -- sharded frame lists, two engines alternating, the record rings, the ONE end-of-job gather
inside the timed region, max-over-ranks time, the single rank-0 line -- i.e. what `python bench.py --gpus N` runs per rank."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubEngine:
    """The attributes and methods bench.run_rank touches, with a deterministic stand-in for the frame."""
    B = 1

    def __init__(self, dev):
        self.dev = dev
        self.out = dict(count=torch.zeros((1,), dtype=torch.int32))
        self.err = torch.zeros((1,), dtype=torch.int32)
        self.records = None
        self.tile_cfg, self.sparse_split, self.sk_ws, self.sk_workgroups = {}, {}, None, 0
        self.post_max = 100
        self.frames = 0

    def set_points(self, pts):
        self._n = int(pts[0].shape[0])

    def adopt_tuning(self, other):
        pass

    def enqueue(self):
        n = self._n % 7
        self.out["count"][0] = n
        self.frames += 1
        if self.records is not None:
            c = int(self.record_cursor.item())
            slot = c % self.records.shape[0]
            self.records[slot].zero_()
            self.records[slot, :n, 7] = 0.5
            self.record_counts[slot] = n
            self.record_cursor += 1

    replay = enqueue

    def capture(self):
        pass

    def attach_records(self, cap):
        self.records = torch.zeros((cap, self.post_max, 9))
        self.record_counts = torch.zeros((cap,), dtype=torch.int32)
        self.record_cursor = torch.zeros((1,), dtype=torch.int32)

    def results(self):
        n = int(self.out["count"][0])
        return [dict(box3d_lidar=np.zeros((n, 7), np.float32), scores=np.full((n,), 0.5, np.float32), label_preds=np.zeros((n,), np.int64))]


def _factory(args, dev):
    return None, [StubEngine(dev) for _ in range(max(1, args.streams))]


def _worker(rank, world, port, argv, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    args = bench.parse(argv)
    out = bench.run_rank(args, rank, world, rank, backend="gloo", device="cpu", engine_factory=_factory)
    q.put((rank, out))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rank_function_two_gloo_ranks_on_a_stub_engine(capfd):
    world, steps = 2, 6
    argv = ["--gpus", str(world), "--steps", str(steps), "--warmup", "2", "--eager", "--no-autotune", "--no-roofline", "--no-host-io",
            "--cpu-frames", "0", "--points", "300", "--pool", "4"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None  # one line, from rank 0
    out = res[0]
    assert out["n_gpus"] == world and out["steps"] == steps and out["scaling"] == "weak"
    assert out["config"]["records_gathered"] == world * steps     # every timed frame of every rank reached the gather
    assert out["config"]["rccl_ranks_seen"] == world
    assert out["config"]["frames_in_flight"] == 4 and out["config"]["cu_sets"] == 0 and out["cu_sets"] is None   # (the default; CU-masked streams exist on the GPU only)
    assert all(not isinstance(v, (dict, list)) for v in out["config"].values())   # scalars only: what the driver's record keeps
    assert out["config"]["collective_backend"] == "gloo" and out["config"]["collectives_in_timed_region"] >= 2 * 4 + 1
    assert out["value"] > 0 and abs(out["value"] - world * steps / (out["ms_per_step"] * 1e-3 * steps)) < 1e-6 * out["value"]
    assert "parity" not in out  # off the GPU (stub engines) there is no gate; on GPUs every rank runs a short one (bench.py)
    json.dumps(out)


def test_single_rank_no_process_group():
    sys.path[:0] = [ROOT]
    import bench
    args = bench.parse(["--steps", "5", "--warmup", "1", "--eager", "--no-autotune", "--no-roofline", "--no-host-io", "--cpu-frames", "0",
                        "--points", "300", "--pool", "4", "--streams", "1"])
    out = bench.run_rank(args, 0, 1, 0, backend="gloo", device="cpu", engine_factory=_factory)
    assert out["n_gpus"] == 1 and out["config"]["records_gathered"] == 5 and "parity" not in out


def test_help_text_formats(capsys):
    """argparse %-formats every help string: a bare per-cent sign in one of them broke `bench.py --help` for two rounds"""
    sys.path[:0] = [ROOT]
    import bench
    import pytest
    with pytest.raises(SystemExit) as ex:
        bench.parse(["--help"])
    assert ex.value.code == 0 and "--cu-split" in capsys.readouterr().out
