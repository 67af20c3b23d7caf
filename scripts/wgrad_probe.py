"""Weight gradient of one 3x3 stride-1 SSFA layer: direct pixel-reduction kernel vs the Winograd-domain kernel, back-to-back launches
on random maps (batch 4: the training step's shapes). Prints us per call (partial + reduce) and the executed / algorithmic rate."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "se-ssd_amd"))
import torch
from sessd_hip import ops

dev = torch.device("cuda:0")
out = []
SHAPES = [(4, 128, 200, 176), (4, 256, 100, 88), (1, 128, 200, 176)]
for (B, C, H, W) in ([SHAPES[int(a)] for a in sys.argv[1:]] or SHAPES):   # optional arguments: indices into SHAPES
    x = torch.randn(B, C, H, W, device=dev)
    g = torch.randn(B, C, H, W, device=dev)
    row = {"shape": [B, C, H, W]}
    for name, wino in (("direct", False), ("winograd", True)):
        for _ in range(3):
            ops.conv2d_wgrad(x, g, 3, 1, winograd=wino)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d_wgrad(x, g, 3, 1, winograd=wino)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        flop = 2.0 * B * H * W * C * C * 9
        row[name] = {"us": round(us, 1), "algorithmic_TFLOPs": round(flop / us / 1e6, 1)}
    out.append(row)
print(json.dumps(out))
