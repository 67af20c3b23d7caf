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
ref = None
for cfg in (4, 12, 20):
    out = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg)
    torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    err = float((out - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("cfg %d: %.3f ms  %.1f TF  maxdiff %.2e" % (cfg, ms, 10.38 / ms, err), flush=True)
