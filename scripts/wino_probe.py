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
cfgs = [int(c) for c in sys.argv[1:]] or [20]
for cfg in cfgs:
    out = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg)
    for _ in range(10):
        ops.conv2d(x, pc, None, None, False, out=out, tile_cfg=cfg)
    torch.cuda.synchronize()
