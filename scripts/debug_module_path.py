import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np, torch
from sessd_hip import configs, synth, ops
from sessd_hip.engine import InferenceEngine
from sessd_hip.anchors import create_anchors_3d_range
dev = torch.device("cuda:0"); VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
frame = synth.make_frame(11, 20000); pts = torch.from_numpy(frame).to(dev)
r = ops.voxelize_batch([pts], VG["voxel_size"], VG["range"], 5, 16000); m = int(r["prefix"][1].item())
eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
eng.set_points([pts]); eng.enqueue(); ref = eng.results()[0]
with torch.no_grad():
    feat = model.reader(r["voxels"][:m], r["num_points"][:m])
    print("vfe diff", float((feat - eng.vfeat[:m]).abs().max()))
    bev = model.backbone(feat, r["coors"][:m], 1, [1408, 1600, 40])
    print("bev diff", float((bev - eng.bev).abs().max()), float(eng.bev.abs().max()))
    x = model.neck(bev)
    print("ssfa diff", float((x - eng.t["out"]).abs().max()), float(eng.t["out"].abs().max()))
    x2 = model.neck(eng.bev)
    print("ssfa(engine bev) diff", float((x2 - eng.t["out"]).abs().max()))
    p = model.bbox_head(x)[0]
    print("head diff", float((p["_planar"].reshape(1, 22, -1) - eng.head).abs().max()), float(eng.head.abs().max()))
