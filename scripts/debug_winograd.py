import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch, torch.nn.functional as F
from sessd_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (cin, cout, H, W) in [(8, 32, 8, 8), (16, 32, 8, 128), (128, 128, 24, 40)]:
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), padding=1)
    pc = ops.pack_conv2d(w.to(dev), 1)
    out = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=20).cpu().double()
    err = (out - ref).abs()
    print(cin, cout, H, W, "max err", float(err.max()), "max ref", float(ref.abs().max()))
    bad = (err > 1e-3).nonzero()
    print("  bad count", len(bad), "of", err.numel())
    if len(bad):
        print("  bad couts", sorted(set(bad[:, 1].tolist()))[:40])
        print("  bad ys", sorted(set(bad[:, 2].tolist()))[:40])
        print("  bad xs", sorted(set(bad[:, 3].tolist()))[:40])
        b0 = bad[0]
        print("  first", b0.tolist(), float(out[tuple(b0)]), float(ref[tuple(b0)]))
