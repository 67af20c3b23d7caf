import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch
from sessd_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(dev)
pc = ops.pack_conv2d(w, 1)
x = torch.randn(1, 128, 200, 176, generator=g).to(dev)
cfgs = [c for c in sys.argv[1:]] or ["20"]
for c in cfgs:   # "22" or "22:248" = stream-K with 248 workgroups; "22:64:100x88" = on a 100x88 map; "22:0:200x176:4:128" = batch 4, 128 channels
    f = c.split(":")
    cfg, wg = int(f[0]), (int(f[1]) if len(f) > 1 and f[1] else 0)
    if len(f) > 2 and f[2]:
        hh, ww = [int(v) for v in f[2].split("x")]
        bb = int(f[3]) if len(f) > 3 else 1
        cc = int(f[4]) if len(f) > 4 else 128
        if cc != w.shape[0]:
            w = (torch.randn(cc, cc, 3, 3, generator=g) * 0.03).to(dev)
            pc = ops.pack_conv2d(w, 1)
        x = torch.randn(bb, cc, hh, ww, generator=g).to(dev)
    out = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg, workgroups=wg)
    for _ in range(10):
        ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg, workgroups=wg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg, workgroups=wg)
    e1.record()
    torch.cuda.synchronize()
    print("cfg %s: %.1f us per launch (back to back)" % (c, e0.elapsed_time(e1) * 1000 / 50))
