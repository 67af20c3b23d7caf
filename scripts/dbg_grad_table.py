"""Whole-detector gradients vs the CPU oracle (the body of tests/test_train_gpu.py::test_whole_model_gradients_vs_oracle): prints
max-abs / max and the relative L2 error of every parameter tensor above 5e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_train_gpu as T
dev = torch.device("cuda:0")
model = T.configs.build_synthetic_detector(dev, seed=0)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
model.train()
frames, ex = T._example(dev, (41, 42), 8000, 8000)
loss = T._loss(model.forward_preds(ex)); loss.backward()
ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
feats, coors = [], []
for b, pts in enumerate(frames):
    v, c, n = T.capi.points_to_voxel(pts, T.VG["voxel_size"], T.VG["range"], 5, 8000)
    feats.append(T.capi.vfe_mean(v, n, 4)); coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
convs = [ref["backbone.middle_conv.%d.weight" % (3 * i)] for i in range(14)]
bns = [{k: ref["backbone.middle_conv.%d.%s" % (3 * i + 1, k)] for k in ("weight", "bias", "running_mean", "running_var")} for i in range(14)]
bev = T.osc.spmiddle_fhd(torch.from_numpy(np.concatenate(feats, 0)), np.concatenate(coors, 0), 2, [1408, 1600, 40], convs, bns, training=True)
lr = T._loss(T.dense_head.head_forward(T.dense_head.ssfa_forward(bev, ref, training=True), ref)); lr.backward()
print("loss", float(loss), float(lr))
for name, p in model.named_parameters():
    want = ref[name].grad; d = p.grad.cpu() - want
    e, s2 = float(d.abs().max()) / float(want.abs().max()), float(d.double().norm()) / float(want.double().norm())
    if e > 5e-4:
        print("%-40s max %.2e  l2 %.2e  shape %s" % (name, e, s2, tuple(p.shape)))
