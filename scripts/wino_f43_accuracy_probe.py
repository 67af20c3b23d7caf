"""CPU arithmetic experiment (the GPU only builds / calibrates the seeded detector): would Winograd F(4x4,3x3) in float32 keep the SSFA neck + heads inside the parity tolerance?
The seven 3x3 stride-1 convs of the neck are replaced by a float32 emulation of F(m x m, 3x3) (m = 2: what the HIP kernels do;
m = 4: 2.25 -> 4 multiply saving, i.e. 1.78x fewer MFMAs) and the head outputs are compared with a float64 run of the same
network on the same BEV tensor (seeded synthetic detector, 20 k-point synthetic frame). Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import dense_head, pipeline, postprocess as pp
from sessd_hip import configs, synth

torch.set_num_threads(16)
MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                  [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_conv(x, w, m):
    """float32 F(m x m, 3x3), padding 1: V = B^T d B per tile, M = sum_c U V, Y = A^T M A (every product / sum in float32)"""
    Bt, G, At = (torch.from_numpy(a).float() for a in MATS[m])
    n = m + 2
    B_, C, H, W = x.shape
    assert H % m == 0 and W % m == 0
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, n, m).unfold(3, n, m)                        # (B, C, th, tw, n, n)
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, d, Bt)            # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                  # G g G^T (the host packs this once, in float64 -> float32)
    U = torch.einsum("ij,ocjk,lk->ocil", G.double(), w.double(), G.double()).float()
    M = torch.einsum("ocil,bcyxil->boyxil", U, V)
    Y = torch.einsum("ij,boyxjk,lk->boyxil", At, M, At)            # (B, O, th, tw, m, m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], H, W)


def run(bev, sd, mode):
    orig = F.conv2d

    def conv2d(x, w, b=None, stride=1, padding=0, **kw):
        if mode and w.shape[-1] == 3 and stride == 1 and w.shape[1] >= 128:
            return wino_conv(x, w, mode)
        return orig(x, w, b, stride=stride, padding=padding, **kw)

    dense_head.F.conv2d = conv2d
    try:
        with torch.no_grad():
            return dense_head.head_forward(dense_head.ssfa_forward(bev, sd), sd)
    finally:
        dense_head.F.conv2d = orig


dev = torch.device("cuda:0")   # build_synthetic_detector calibrates its BatchNorms through the device modules
model = configs.build_synthetic_detector(dev, seed=0)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
VG = configs.VOXEL_GENERATOR
frame = synth.make_frame(1, 20000)
anchors = pp.create_anchors_3d_range().reshape(-1, 7)
_, inter = pipeline.run_frames([frame], sd, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
bev = inter["bev"].float()
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
want = run(bev.double(), sd64, 0)
out = {}
for name, mode in (("direct_f32", 0), ("winograd_F2x2_f32", 2), ("winograd_F4x4_f32", 4)):
    got = run(bev, sd, mode)
    r = {}
    for k in want:
        e = (got[k].double() - want[k]).abs()
        r[k] = {"max_abs_err": float(e.max()), "max_abs_value": float(want[k].abs().max()),
                "p99.9_abs_err": float(torch.quantile(e.flatten()[:: max(1, e.numel() // 1000000)], 0.999))}
    # the anchors that reach the post-processor (score >= 0.3): their box codes are what the detection tolerance is about
    # (centre = code * anchor diagonal (4.2 m) + anchor centre, size = exp(code) * anchor size; box_tol 2e-3)
    keep = torch.sigmoid(want["cls_preds"].reshape(-1)) >= 0.3
    eb = (got["box_preds"].double() - want["box_preds"]).abs().reshape(-1, 7)[keep]
    r["box_codes_of_anchors_over_threshold"] = {"n": int(keep.sum()), "max_abs_err": float(eb.max()) if len(eb) else 0.0,
                                                "max_abs_code": float(want["box_preds"].reshape(-1, 7)[keep].abs().max()) if len(eb) else 0.0}
    out[name] = r
print(json.dumps(out, indent=1))
