"""Times sessd_conv2d_wgrad on the stride-2 3x3 layer of the SSFA neck at the training batch (4 x 128 -> 256, 200x176 -> 100x88):
SESSD_WGRAD_S2_LDS=0 (private-operand kernel of round 3) vs the default (LDS-staged kernel of round 4); the 1x1 layers for
reference. Run once per setting (the switch is read once per process). Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch

from sessd_hip import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
out = {"s2_lds": os.environ.get("SESSD_WGRAD_S2_LDS", "1")}
for name, (ci, co, k, s, H, W) in {"b1.0 3x3 s2 128->256 @200x176": (128, 256, 3, 2, 200, 176),
                                    "trans_0 1x1 128->128 @200x176": (128, 128, 1, 1, 200, 176),
                                    "trans_1 1x1 256->256 @100x88": (256, 256, 1, 1, 100, 88)}.items():
    x = torch.randn(4, ci, H, W, generator=g).to(dev)
    gy = torch.randn(4, co, H // s, W // s, generator=g).to(dev)
    for _ in range(3):
        gw = ops.conv2d_wgrad(x, gy, k, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gw = ops.conv2d_wgrad(x, gy, k, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flop = 2.0 * 4 * co * ci * k * k * (H // s) * (W // s)
    out[name] = {"us_partial_plus_reduce": us, "tflops": flop / us / 1e6, "checksum": float(gw.double().abs().sum())}
print(json.dumps(out))
