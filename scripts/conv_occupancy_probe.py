import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch
from sessd_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(dev)
pc = ops.pack_conv2d(w, 1)
for cfg, pxwg in ((1, 64), (3, 32), (2, 64)):
    for nwg in (64, 128, 256, 384, 512, 550, 640, 768, 1024, 1100, 1280, 2048):
        H, W = nwg, pxwg
        x = torch.randn(1, 128, H, W, generator=g).to(dev)
        out = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * H * W * 128 * 128 * 9
        print("cfg %d nWG %5d: %.3f ms  %.1f TF   (%.1f us per WG-wave-of-256)" % (cfg, nwg, ms, fl / ms / 1e9, ms * 1e3 / max(1, (nwg + 255) // 256)), flush=True)
