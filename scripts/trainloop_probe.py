"""Sustained rate of the training loop (sessd_hip/trainloop.fit) with the data path on one stream / overlapped on a side stream,
beside the replay of one resident batch and the data path alone. One MI355X."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
import torch

from sessd_hip import configs, trainloop
from sessd_hip import train as strain

dev = torch.device("cuda:0")
pool = trainloop.ScenePool(range(700, 724), 20000)
out = {}
N = 120
for overlap in (False, True):
    model = configs.build_synthetic_detector(dev, seed=0)
    step, rep = trainloop.fit(model, pool, iterations=N, batch=4, seed=1, log_every=N, overlap=overlap)
    out["fit_overlap_%s_ms_per_iteration" % overlap] = rep["ms_per_iteration"]
    if overlap:
        # the same graph on a resident batch, and the loader alone
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            step.replay()
        torch.cuda.synchronize()
        out["replay_only_ms"] = (time.perf_counter() - t0) / 40 * 1e3
        data = trainloop.DeviceBatcher(pool, dev, 4, 60, seed=2)
        data.load(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(1, 41):
            data.load(it)
        t_host = (time.perf_counter() - t0) / 40 * 1e3
        torch.cuda.synchronize()
        out["loader_only_ms_gpu_complete"] = (time.perf_counter() - t0) / 40 * 1e3
        out["loader_only_ms_host_enqueue"] = t_host
    step.graph = None
print(json.dumps(out, indent=1))
