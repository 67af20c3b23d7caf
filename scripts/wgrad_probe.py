import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch
from sessd_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, ci, co, H, W, k, s) in ((4, 128, 128, 200, 176, 3, 1), (4, 256, 256, 100, 88, 3, 1), (4, 128, 256, 200, 176, 3, 2), (1, 128, 128, 200, 176, 3, 1)):
    x = torch.randn(B, ci, H, W, generator=g).to(dev)
    gy = torch.randn(B, co, H // s, W // s, generator=g).to(dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, gy, k, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_wgrad(x, gy, k, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * (H // s) * (W // s) * ci * co * k * k
    print("wgrad B%d %d->%d %dx%d k%d s%d: %.3f ms %.1f TF" % (B, ci, co, H, W, k, s, ms, fl / ms / 1e9), flush=True)
