"""tile_cfg 22 (second generation: M exchange through LDS) vs 24 (third: output transform in registers) on the SSFA shapes,
back-to-back launches on a hot input, HIP events. One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch

from sessd_hip import ops

dev = torch.device("cuda:0")
out = {}
for name, (B, C, H, W) in {"128x128@200x176": (1, 128, 200, 176), "256x256@100x88": (1, 256, 100, 88), "128x128@200x176 b2": (2, 128, 200, 176),
                           "128x128@200x176 b4": (4, 128, 200, 176)}.items():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    pc = ops.pack_conv2d(w, 1)
    y = torch.empty_like(x)
    row = {}
    for cfg in (22, 23, 24):
        ws = ops.winograd_sk_workspace(B, H, W, C, dev, 0, cfg - 22)
        for _ in range(5):
            ops.conv2d(x, pc, sc, sh, True, None, y, cfg, workspace=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.conv2d(x, pc, sc, sh, True, None, y, cfg, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        row[str(cfg)] = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
    out[name] = row
print(json.dumps(out))
