"""Run the inference engine eagerly for a few frames (for rocprofv3 --kernel-trace / --pmc passes of the sparse stage).

"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    sys.path.insert(0, p)
import torch
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine

ap = argparse.ArgumentParser()
ap.add_argument("--stress", action="store_true")
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--sort", action="store_true")
ap.add_argument("--graph", action="store_true")
ap.add_argument("--force-active", action="store_true",
                help="blocks 0 / 1 of the neck in active-tile mode whatever the autotune timed (under a counter pass every launch "
                     "carries the profiler's overhead and the autotune declines the two extra launches)")
ap.add_argument("--cu-half", action="store_true",
                help="run the engine as bench.py's default runs its four: on a CU-masked stream over one half of the chip, persistent "
                     "launches sized for 128 CUs, autotuned there")
ap.add_argument("--fixed", action="store_true",
                help="NO autotune: engine.force_active_tiles() (the configuration the autotune ends in on MI355X, as smoke() runs it) -- every "
                     "counter pass of a series then profiles the SAME launches (separate passes of an autotuned probe can pick different "
                     "tilings: round 6 found FETCH / WRITE columns that belonged to different kernels)")
ap.add_argument("--list-shares", default="auto", choices=["auto", "whole", "cut"],
                help="with --force-active: the Winograd list layers on whole-unit shares (round 5) / stream-K shares")
a = ap.parse_args()
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
pts, mv, B, ss = (200000, 64000, 8, 3) if a.stress else (20000, 16000, 1, 1)
model = configs.build_synthetic_detector(dev, seed=0, max_voxels=mv, num_points=pts, supersample=ss)
frames = [torch.from_numpy(synth.make_frame(i, pts, supersample=ss)).to(dev) for i in range(B)]
kw = {}
if a.sort:
    kw["sort_sites"] = True
e = InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], mv, configs.TEST_CFG, batch_size=B,
                    max_points_per_frame=pts, device=dev, **kw)
e.set_points(frames)
e.enqueue()
torch.cuda.synchronize()
if a.cu_half:
    from sessd_hip import ops as _ops
    _st, _ncu = _ops.cu_masked_stream(0, 2, dev)
    e.cu_budget = _ncu
    torch.cuda.set_stream(_st)
if a.fixed:
    e.force_active_tiles()
    e.set_list_shares("whole" if a.list_shares == "auto" else a.list_shares)
    print("fixed configuration:", e.tile_cfg, e.active_cfg)
else:
    e.autotune()
if a.force_active and not a.fixed and e.ta is not None:
    from sessd_hip import ops
    need = max(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(2 * B, e.H, e.W, 256, 1, 0)),
               int(ops.lib.sessd_conv2d_sk_workspace_bytes(B, e.H, e.W, 256, 1, 0)))
    if e.sk_ws is None or e.sk_ws.numel() < need:
        e.sk_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    e.active_cfg = {0: (1, 4), 1: (1, 8), 2: (1, 16), 3: (30, 4), 4: (1, 4), 5: (1, 8), 6: (4, 0), 8: (3, 0)}
    e.set_list_shares(a.list_shares)
    print("active_tiles forced:", e.active_cfg)
if a.graph:
    e.capture()
for i in range(a.frames):
    e.set_points(frames)
    if a.graph:
        e.replay()
    else:
        e.enqueue()
torch.cuda.synchronize()
print("sites", e.spmiddle_algorithmic_bytes())
print("active_tile_fractions", {k: round(v, 4) for k, v in e.active_tile_fractions().items()})
print("stages", e.stage_times(reps=5))
if not a.fixed:
    print("tuning", {k: (v[0] & 255, v[0] >> 8, round(v[1] * 1e3, 1)) for k, v in e.tune_report.items() if k.startswith("sparse")})
