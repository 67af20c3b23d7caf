"""bench.py at N > 1 on real engines (round-4 review: "no N > 1 result is ever held to the oracle; every rank autotunes on its
own"): two ranks -- two processes on the one GPU of the test box, collectives over gloo, as the two-rank training tests do -- each
with its own autotuned, captured engines. Every rank holds its engines to the CPU oracle on a short sample of ITS frames before
the clock starts; the verdicts are summed over the ranks and rank 0's line carries them (`parity.ranks`, also inside `config`)."""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, argv, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import bench
        args = bench.parse(argv)
        out = bench.run_rank(args, rank, world, 0, backend="gloo")   # both ranks on cuda:0
        q.put((rank, out))
    except SystemExit as ex:
        q.put((rank, {"exit": str(ex)}))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()[-1500:]}))


@pytest.mark.parametrize("streams", [1, 4])
def test_two_ranks_each_hold_their_engines_to_the_oracle(streams):
    """streams = 4: bench.py's default since round 5 -- every rank builds four engines on CU-masked streams of its device (two CU
    sets); here both ranks share the one GPU, so eight engines run on the two halves"""
    world, steps = 2, 8
    argv = ["--gpus", str(world), "--steps", str(steps), "--warmup", "2", "--streams", str(streams), "--cpu-frames", "3", "--no-roofline",
            "--no-host-io", "--no-sequential", "--no-train-step", "--pool", "4", "--spinup-seconds", "0.1"]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        # run_rank's ordered teardown (graphs and engines, allocator cache, process group, then the CU-masked streams) must leave
        # a process that exits cleanly
        assert p.exitcode == 0, p.exitcode
    assert res[1] is None, res[1]
    out = res[0]
    assert "error" not in out and "exit" not in out, out
    par = out["parity"]
    assert par["ranks"] == 2 and par["frames"] == 2 * 3 * streams and par["matched"] == par["frames"] and par["ok"], par
    cfg = out["config"]
    assert all(not isinstance(v, (dict, list)) for v in cfg.values())   # scalars only: what the driver's record keeps
    assert cfg["parity_ok"] and cfg["parity_frames"] == 6 * streams and cfg["parity_matched"] == 6 * streams and cfg["parity_rule"] == "synthetic"
    assert cfg["frames_in_flight"] == streams and (cfg["cu_sets"] == 2) == (streams > 1) and (out["cu_sets"] is not None) == (streams > 1)
    assert cfg["weights"] == "random"   # (--no-train-step: nothing trains, the round-5 line)
    assert out["n_gpus"] == 2 and out["config"]["records_gathered"] == world * steps and out["config"]["rccl_ranks_seen"] == 2
    assert "cpu_baseline" not in out   # the baseline figure stays on the N = 1 line
    json.dumps(out)


def test_two_ranks_train_first_and_gate_strictly():
    """the driver's command at N > 1 (`bench.py --gpus N --steps K --warmup W`, no other flag): every rank trains the detector in its
    own process BEFORE the process group exists (the captured iteration of a one-rank job: no SyncBN, one graph), then joins the
    group, builds its engines on CU-masked streams, and gates them under the STRICT rule on a sample of its own frames"""
    world, steps = 2, 8
    argv = ["--gpus", str(world), "--steps", str(steps), "--warmup", "2", "--cpu-frames", "3", "--no-roofline", "--no-host-io", "--no-sequential",
            "--pool", "4", "--spinup-seconds", "0.1", "--train-replays", "2"]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, p.exitcode
    out = res[0]
    assert res[1] is None and "error" not in out and "exit" not in out, (res[1], out)
    cfg, par = out["config"], out["parity"]
    assert cfg["weights"] == "trained_in_process" and cfg["parity_rule"] == "strict" and cfg["parity_ok"], cfg
    assert par["ranks"] == 2 and par["frames"] == 2 * 3 * 4 and par["matched"] == par["frames"], par
    assert out["train_step"]["pretrain_iterations"] == 300 and out["train_step"]["sparse_overflow_flag"] == 0
    assert cfg["frames_in_flight"] == 4 and cfg["cu_sets"] == 2 and cfg["records_gathered"] == world * steps and cfg["rccl_ranks_seen"] == 2
