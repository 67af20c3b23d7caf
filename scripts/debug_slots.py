import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np, torch
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
for (P, MV, ss) in ((20000, 16000, 1), (200000, 64000, 3)):
    fa, fb = synth.make_frame(100, P, supersample=ss), synth.make_frame(101, P, supersample=ss)
    frames = [fa, fb, fa, fb]
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=4, max_points_per_frame=P, device=dev)
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames])
    eng.enqueue()
    got = eng.results()
    print("P", P, "prefix", eng.prefix.cpu().numpy(), "levels", [int(l["n"].item()) for l in eng.levels[1:]], "dets", [len(g["scores"]) for g in got])
    bev = eng.bev
    print(" bev slot0==slot2", bool(torch.equal(bev[0], bev[2])), float((bev[0] - bev[2]).abs().max()), "slot1==slot3", bool(torch.equal(bev[1], bev[3])))
    for name in eng.t:
        t = eng.t[name]
        if t.dim() == 4 and t.shape[0] == 4:
            print("  ", name, tuple(t.shape), bool(torch.equal(t[0], t[2])), float((t[0] - t[2]).abs().max()))
    for k in ("box3d_lidar", "scores", "label_preds"):
        print(" ", k, np.array_equal(got[0][k], got[2][k]), np.array_equal(got[1][k], got[3][k]))
    if not np.array_equal(got[0]["scores"], got[2]["scores"]):
        print(got[0]["scores"][:10], got[2]["scores"][:10])
