"""Which detection differs between the engine and the oracle on the dense-scene frame, and why (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import pipeline, postprocess as pp, capi
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
P, MV = 200000, 64000
model = configs.build_synthetic_detector(dev, seed=0, calib_frame_seed=99, max_voxels=MV, num_points=P, supersample=3)
state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
frame = synth.make_frame(100, P, supersample=3)
anchors = pp.create_anchors_3d_range().reshape(-1, 7)
B = int(os.environ.get("B", "8"))
distinct = [synth.make_frame(100 + i, P, supersample=3) for i in range(3)]
frames = [distinct[i % 3] for i in range(B)]
eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=B, max_points_per_frame=P, device=dev)
eng.set_points([torch.from_numpy(f).to(dev) for f in frames]); eng.enqueue()
got = eng.results()[0]
print("device boxes", got["box3d_lidar"])
want, inter = pipeline.run_frames([frame], state, VG["range"], VG["voxel_size"], 5, MV, anchors, None, return_intermediate=True)
dbg = inter["debug"][0]
print("device", len(got["scores"]), got["scores"], "\noracle", len(want[0]["scores"]), want[0]["scores"])
kb = dbg["cand_boxes"][dbg["nms_kept_rows"]]
pr = np.array([0, -40.0, -5.0, 70.4, 40.0, 5.0], np.float32)
dist = np.minimum(np.abs(kb[:, :3] - pr[:3]), np.abs(kb[:, :3] - pr[3:])).min(1)
print("oracle box", want[0]["box3d_lidar"])
print("kept boxes closest to a range face:", [(int(r), kb[i, :3].tolist(), float(dist[i])) for i, r in enumerate(dbg["nms_kept_rows"]) if dist[i] < 0.05])
print("candidates", dbg["num_candidates"], "topk", dbg["topk"], "near", dbg["near_pairs"].tolist(), "kept rows", dbg["nms_kept_rows"].tolist())
cand = dbg["cand_boxes"]; cs = dbg["cand_dets"][:, 5]
for i, (b, s) in enumerate(zip(got["box3d_lidar"], got["scores"])):
    d = np.hypot(cand[:, 0] - b[0], cand[:, 1] - b[1])
    j = int(np.argmin(d))
    print("device det", i, "score %.6f" % s, "nearest candidate row", j, "dist %.2e" % d[j], "cand score %.6f" % cs[j], "kept by oracle:", j in set(dbg["nms_kept_rows"].tolist()))
    if j not in set(dbg["nms_kept_rows"].tolist()):
        corners = capi.box2d_corners(dbg["cand_dets"])
        for k in dbg["nms_kept_rows"].tolist():
            if k < j:
                iou = capi.quad_iou(corners[k], corners[j])
                if iou > 0:
                    print("   overlaps kept row", k, "IoU %.6f" % iou)
# head-level difference: scores near the threshold / BEV error
head = eng.head.cpu().numpy()[0]  # (22, H*W)
cls_dev = head[14:16].reshape(2, -1).T.reshape(-1)  # anchor-major? check against oracle preds
cls_or = inter["preds"]["cls_preds"][0].reshape(-1).numpy()
print("max |cls logit diff| (pixel-major a fastest):", float(np.abs(head[14:16].T.reshape(-1) - cls_or).max()))
s_or = pp.sigmoid32(cls_or)
print("oracle scores within 1e-4 of 0.3:", int((np.abs(s_or - 0.3) < 1e-4).sum()))
