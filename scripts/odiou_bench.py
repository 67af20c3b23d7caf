"""Device time of the ODIoU op (value + gradient) and the CPU oracle beside it; one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np
import torch
from sessd_hip import ops
from oracle import odiou
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
out = {}
for n in (256, 4096):
    g = np.zeros((n, 7), np.float32)
    g[:, 0] = rng.uniform(0, 60, n); g[:, 1] = rng.uniform(-30, 30, n); g[:, 2] = -1
    g[:, 3] = 1.6; g[:, 4] = 3.9; g[:, 5] = 1.5; g[:, 6] = rng.uniform(-3, 3, n)
    q = (g + rng.normal(0, 0.2, (n, 7))).astype(np.float32)
    gt, w = torch.from_numpy(g).to(dev), torch.ones(n, device=dev)
    qt = torch.from_numpy(q).to(dev).requires_grad_(True)
    for _ in range(3):
        ops.odiou_3d_loss(gt, qt, w, 4).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        qt.grad = None
        ops.odiou_3d_loss(gt, qt, w, 4).backward()
    torch.cuda.synchronize()
    out["device_ms_n%d" % n] = (time.perf_counter() - t0) / 20 * 1e3
c0 = time.perf_counter()
odiou.odiou_loss(g[:256], q[:256], np.ones(256), 4)
out["cpu_oracle_ms_n256"] = (time.perf_counter() - c0) * 1e3
print(json.dumps(out))
