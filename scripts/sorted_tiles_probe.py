"""Per sparse layer: the plain 16-row tiles against the offset-pattern tiles (InferenceEngine(sort_tiles=True)) with the layer's
tuned launch configuration, back-to-back launches, HIP events -- batch 1 and the dense-scene batch. One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch

from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine

dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
out = {}
for name, (B, P, MV, ss) in {"batch1": (1, 20000, 16000, 1), "stress": (8, 200000, 64000, 3)}.items():
    model = configs.build_synthetic_detector(dev, seed=0, calib_frame_seed=99 if ss == 3 else 0, max_voxels=MV, num_points=P, supersample=ss)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, B, P, dev, sort_tiles=True)
    eng.set_points([torch.from_numpy(synth.make_frame(100 + i, P, supersample=ss)).to(dev) for i in range(B)])
    eng.enqueue()
    torch.cuda.synchronize()
    eng.autotune()
    rows = []
    for srt in (False, True):
        for k in list(eng.sparse_sorted):
            eng.sparse_sorted[k] = srt
        eng.sparse_sorted[13] = srt
        rep = eng.spmiddle_mfma_report(reps=20)
        rows.append(rep)
    out[name] = {"layers": [dict(layer=a["layer"], cin=a["cin"], cout=a["cout"], sites=a["sites"], plain_ms=a["ms"], sorted_ms=b["ms"],
                                 plain_steps=a["tile_steps"], sorted_steps=b["tile_steps"]) for a, b in zip(rows[0]["layers"], rows[1]["layers"])],
                 "plain_conv_ms": rows[0]["conv_ms"], "sorted_conv_ms": rows[1]["conv_ms"],
                 "plain_useful": rows[0]["useful_row_fraction"], "sorted_useful": rows[1]["useful_row_fraction"],
                 "tuning": {str(k): v for k, v in eng.sparse_split.items()}}
    del eng, model
print(json.dumps(out))
