"""Host time of one frame's submission (set_points + graph replay) against its device time: is the timed region host-bound with
several frames in flight? One MI355X."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
import torch

from sessd_hip import configs, ops, synth
from sessd_hip.engine import InferenceEngine

dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
frames = [torch.from_numpy(synth.make_frame(i, 20000)).to(dev) for i in range(8)]
engines, streams = [], []
for k in range(4):
    st, ncu = ops.cu_masked_stream(k % 2, 2, dev)
    e = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    e.cu_budget = ncu
    e.set_points([frames[0]])
    if k == 0:
        with torch.cuda.stream(st):
            e.enqueue(); torch.cuda.synchronize(); e.autotune()
    else:
        e.adopt_tuning(engines[0])
    e.attach_records(4096)
    with torch.cuda.stream(st):
        e.capture()
    engines.append(e); streams.append(st)
torch.cuda.synchronize()
out = {}


def drive(idx, n, host_acc):
    t_host = 0.0
    for i in range(n):
        k = idx[i % len(idx)]
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[k]):
            engines[k].set_points([frames[i % 8]])
            engines[k].replay()
        t_host += time.perf_counter() - t0
    host_acc.append(t_host / n)


for name, groups in (("one_thread_4_engines", [[0, 1, 2, 3]]), ("two_threads_2_engines_each", [[0, 2], [1, 3]]),
                     ("four_threads_1_engine_each", [[0], [1], [2], [3]]), ("one_thread_1_engine", [[0]])):
    for rep in range(2):
        for e in engines:
            e.record_cursor.zero_()
        torch.cuda.synchronize()
        N = 400
        acc = []
        t0 = time.perf_counter()
        ths = [threading.Thread(target=drive, args=(g, N // len(groups), acc)) for g in groups]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out[name] = {"frames_per_s": N / dt, "host_ms_per_submission": sum(acc) / len(acc) * 1e3, "enqueue_wall_ms_per_frame": t_enq / N * 1e3,
                 "wall_ms_per_frame": dt / N * 1e3}
print(json.dumps(out, indent=1))
