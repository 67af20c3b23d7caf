"""Timing probe of the LDS-tiled stream-K conv kernel (tile_cfg 30) against the direct kernel on the SSFA layers that are not
3x3 stride 1. Usage: python scripts/csk_probe.py [workgroups ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch  # noqa: E402
from sessd_hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
wgs_list = [int(a) for a in sys.argv[1:]] or [0]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
layers = [("b1.0 3x3 s2 128->256 @200x176", "conv", 128, 256, 3, 2, 200, 176, (0, 1, 3, 11, 13)),
          ("trans_0 1x1 128->128 @200x176", "conv", 128, 128, 1, 1, 200, 176, (3, 4)),
          ("trans_1 1x1 256->256 @100x88", "conv", 256, 256, 1, 1, 100, 88, (3, 4)),
          ("deconv 3x3 s2 256->128 @100x88", "deconv", 256, 128, 3, 2, 100, 88, (4, 40, 41, 42))]
if os.environ.get("CSK_LAYERS"):
    layers = [layers[int(i)] for i in os.environ["CSK_LAYERS"].split(",")]
for name, kind, ci, co, k, st, H, W, cfgs in layers:
    if os.environ.get("CSK_SKIP_DIRECT"):
        cfgs = cfgs[:1]
    x = torch.randn(1, ci, H, W, generator=g).to(dev)
    if kind == "conv":
        w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
        pc = ops.pack_conv2d(w, st)
        Ho, Wo, th, tw, ncls = (H + st - 1) // st, (W + st - 1) // st, (H + st - 1) // st, (W + st - 1) // st, 1
        flops = 2.0 * Ho * Wo * ci * co * k * k
    else:
        w = (torch.randn(ci, co, 3, 3, generator=g) * 0.05).to(dev)
        pc = ops.pack_deconv2d_s2(w)
        Ho, Wo, th, tw, ncls = 2 * H, 2 * W, H, W, 4
        flops = 2.0 * H * W * ci * co * 9
    scale, shift = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    out = torch.empty((1, co, Ho, Wo), device=dev)
    ref = None
    for cfg in cfgs:
        us = timeit(lambda: ops.conv2d(x, pc, scale, shift, True, out=out, tile_cfg=cfg))
        if ref is None:
            ref = out.clone()
        print("%-34s cfg %2d           %7.1f us  %6.1f TFLOP/s" % (name, cfg, us, flops / us / 1e6))
    for wgs in wgs_list:
        ws = ops.conv2d_sk_workspace(1, th, tw, co, ncls, dev, wgs)
        us = timeit(lambda: ops.conv2d(x, pc, scale, shift, True, out=out, tile_cfg=30, workspace=ws, workgroups=wgs))
        err = float((out - ref).abs().max())
        print("%-34s cfg 30 wgs %4d  %7.1f us  %6.1f TFLOP/s  max|diff to direct| %.2e" % (name, wgs, us, flops / us / 1e6, err))
