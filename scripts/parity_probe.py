"""GPU-side half of a parity investigation: saves, for the bench's frame pool, the detections and head outputs of the engine in
its default and its autotuned configuration + the calibrated state_dict, so that the comparison with the oracle (which needs the
calibrated weights) can be studied on a machine without a GPU.  python scripts/parity_probe.py OUT.npz [frames...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
import numpy as np
import torch

from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine

out_path = sys.argv[1]
detail = [int(a) for a in sys.argv[2:]] or [9, 10, 11]
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
save = {"sd__" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
frames = [synth.make_frame(i, 20000) for i in range(16)]
for tag in ("default", "tuned"):
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20000, dev)
    eng.set_points([torch.from_numpy(frames[0]).to(dev)])
    eng.enqueue()
    torch.cuda.synchronize()
    if tag == "tuned":
        eng.autotune()
        save["tile_cfg"] = np.array(sorted((k, str(v)) for k, v in eng.tile_cfg.items()))
    for i, f in enumerate(frames):
        eng.set_points([torch.from_numpy(f).to(dev)])
        eng.enqueue()
        r = eng.results()[0]
        save["%s_box_%d" % (tag, i)] = r["box3d_lidar"]
        save["%s_score_%d" % (tag, i)] = r["scores"]
        if i in detail:
            save["%s_head_%d" % (tag, i)] = eng.head.cpu().numpy()
np.savez_compressed(out_path, **save)
print("saved", out_path, os.path.getsize(out_path) >> 20, "MiB")
