"""Per-step cost of bev_tile_activity_kernel: the step program cut short (1 .. 6 steps) on the BEV occupancy of a synthetic
20 k-point scan, 50 launches per captured graph. Prints one JSON line (us per launch)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
from sessd_hip import ops

dev = torch.device("cuda:0")
H, W = 200, 176
rng = np.random.RandomState(1)
cy, cx = rng.randint(10, H - 10, 40), rng.randint(10, W - 10, 40)
y = np.clip(np.concatenate([rng.normal(cy[i], 3, 60) for i in range(40)] + [rng.randint(0, H, 600)]).astype(int), 0, H - 1)
x = np.clip(np.concatenate([rng.normal(cx[i], 4, 60) for i in range(40)] + [rng.randint(0, W, 600)]).astype(int), 0, W - 1)
idx = np.unique(np.stack([np.zeros_like(y), np.zeros_like(y), y, x], 1), axis=0).astype(np.int32)
di = torch.zeros((4096, 4), dtype=torch.int32, device=dev)
di[:len(idx)] = torch.from_numpy(idx).to(dev)
n = torch.tensor([len(idx)], dtype=torch.int32, device=dev)
out = {"sites": int(len(idx))}
full = [0, 0, 0, 2, 0, 0]
for k in range(1, 7):
    ta = ops.TileActivity(1, H, W, full[:k], dev)
    ta.run(di, n, 4096)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ta.run(di, n, 4096)
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                ta.run(di, n, 4096)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
    torch.cuda.synchronize()
    out["steps_%d" % k] = e0.elapsed_time(e1) / 500 * 1e3
    out["frac_%d" % k] = [float(v) / ((d[0] // 2) * (d[1] // 2)) for v, d in zip(ta.n_list.cpu(), ta.dims)]
print(json.dumps(out))
