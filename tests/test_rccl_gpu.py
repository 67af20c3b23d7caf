"""RCCL executed on the one GPU of the test box (round-5 review, "What's weak 8" / "Next round 4"): a process group of ONE rank over
the `nccl` backend with SESSD_FORCE_COLLECTIVES=1 (sessd_hip.dist.collectives_enabled switches the world-1 short-circuits off), so
that every collective of the N-GPU job runs -- from the CU-masked streams bench.py issues them on -- and the process then goes
through bench.py's ordered teardown and EXITS 0, plainly and under rocprofv3 (round 5 saw exit code 139 from masked streams left
to the runtime under the profiler). Reference: tools/dist_test.py:150-186, det3d/core/utils/dist_utils.py:8-57."""
import json
import os
import shutil
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.update(SESSD_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_collectives_on_masked_streams_and_ordered_teardown():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_masked_probe.py")], env=_env(), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = _last_json(r.stdout)
    assert out["ok"] and out["backend"] == "nccl" and out["masked_streams_closed"] == 4 and out["syncbn_collectives"] > 50, out


BENCH = ["bench.py", "--gpus", "1", "--steps", "12", "--warmup", "4", "--cpu-frames", "2", "--no-train-step", "--no-host-io",
         "--no-roofline", "--no-sequential", "--pool", "4", "--spinup-seconds", "0.1"]


def test_bench_rank_over_nccl_exits_cleanly():
    """bench.py itself: one rank, backend nccl, collectives forced -- the records all_gather of four engines from their CU-masked
    streams, the barriers and the MAX all-reduce inside and around the timed region, then destroy_process_group and the masked
    streams' destruction in that order; the exit code is the driver's `rc`"""
    r = subprocess.run([sys.executable] + BENCH, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = _last_json(r.stdout)
    cfg = out["config"]
    assert cfg["collective_backend"] == "nccl" and cfg["collectives_in_timed_region"] >= 2 * 4 + 1 and cfg["rccl_ranks_seen"] == 1, cfg
    assert cfg["cu_sets"] == 2 and cfg["frames_in_flight"] == 4 and cfg["records_gathered"] == 12 and cfg["parity_ok"], cfg


@pytest.mark.skipif(shutil.which("rocprofv3") is None, reason="rocprofv3 not on PATH")
def test_bench_rank_over_nccl_exits_cleanly_under_rocprofv3(tmp_path):
    """the same process under the profiler's tool library (kernel trace only): the teardown path must not depend on it"""
    env = _env()
    env["TMPDIR"] = str(tmp_path)
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", str(tmp_path / "prof"), "-o", "t", "--", sys.executable] + \
          [os.path.join(ROOT, BENCH[0])] + BENCH[1:]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=str(tmp_path))
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = _last_json(r.stdout)
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["parity_ok"], out["config"]
