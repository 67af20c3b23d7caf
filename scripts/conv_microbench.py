import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch
from sessd_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
for (B, cin, cout, k, s, H, W) in [(1, 128, 128, 3, 1, 200, 176), (2, 128, 128, 3, 1, 200, 176), (1, 256, 256, 3, 1, 100, 88), (1, 128, 256, 3, 2, 200, 176), (1, 128, 128, 1, 1, 200, 176)]:
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.03).to(dev)
    pc = ops.pack_conv2d(w, s)
    fl = 2.0 * B * (H // s) * (W // s) * cin * cout * k * k
    for cfg in cfgs:
        out = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("B%d %dx%d %d->%d k%d s%d cfg %d: %.3f ms %.1f TF" % (B, H, W, cin, cout, k, s, cfg, ms, fl / ms / 1e9), flush=True)
