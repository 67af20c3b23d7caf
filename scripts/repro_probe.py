"""Is the training iteration bit-reproducible from run to run? (round-3 review item 7: test_captured_iteration_equals_eager accepted
parameters "less than one Adam step" apart and blamed torch / MIOpen pieces that have since moved onto the HIP kernels.)
Two eager trainers from the same seed on the same batch: per parameter tensor the largest difference of the GRADIENT after
iteration 1 (before any update noise can be amplified by Adam) and of the parameters after 3 iterations; then graph replay vs
eager the same way. Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import torch

from sessd_hip import configs, ops, trainbench, train as strain

dev = torch.device("cuda:0")
real = "--standin" not in sys.argv


def standin(ex, s, t, w):
    p, q = s[0], t[0]
    M = ops.mean_all
    return (M(p["box_preds"].pow(2)) + M(torch.sigmoid(p["cls_preds"])) + 0.2 * M(p["dir_cls_preds"].pow(2)) + M(p["iou_preds"].abs())
            + w * M((p["cls_preds"] - q["cls_preds"]).pow(2)))


def make():
    return strain.TrainStep(configs.build_synthetic_detector(dev, seed=0), None if real else standin, total_steps=100)


ex, cap = trainbench.labelled_batch(dev, 2, npts=9000, max_voxels=8000)
a, b = make(), make()
names = [n for n, _ in a.student.named_parameters()]


def per_param(x, y, flat):
    out = {}
    for n, p, o in zip(names, flat.params, flat.offsets):
        d = float((x[o:o + p.numel()] - y[o:o + p.numel()]).abs().max())
        if d != 0.0:
            out[n] = d
    return out


la, _, _ = a(cap, device_schedule=True)
lb, _, _ = b(cap, device_schedule=True)
torch.cuda.synchronize()
res = {"loss": "real (sessd_head_loss)" if real else "stand-in", "loss_equal_iter1": float(la) == float(lb),
       "grad_diff_iter1": per_param(a.flat_s.grad, b.flat_s.grad, a.flat_s)}
for _ in range(2):
    a(cap, device_schedule=True)
    b(cap, device_schedule=True)
torch.cuda.synchronize()
res["param_diff_after_3_eager"] = per_param(a.flat_s.data, b.flat_s.data, a.flat_s)
res["teacher_equal_after_3_eager"] = bool(torch.equal(a.flat_t.data, b.flat_t.data))
# graph vs eager: c captures after one eager iteration; d runs eagerly; both then do 3 more
c, d = make(), make()
static = {k: ([t.clone() for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else (v.clone() if torch.is_tensor(v) else v))
          for k, v in cap.items()}
c.capture(static, warmup=1)
d(cap, device_schedule=True)
gl, el = [], []
for _ in range(3):
    gl.append(float(c.replay()))
    el.append(float(d(cap, device_schedule=True)[0]))
torch.cuda.synchronize()
res["graph_vs_eager_losses"] = [gl, el]
res["graph_vs_eager_param_diff"] = per_param(c.flat_s.data, d.flat_s.data, c.flat_s)
res["graph_vs_eager_max"] = float((c.flat_s.data - d.flat_s.data).abs().max())
res["eager_vs_eager_max"] = float((a.flat_s.data - b.flat_s.data).abs().max())
for k in ("grad_diff_iter1", "param_diff_after_3_eager", "graph_vs_eager_param_diff"):
    full = res[k]
    res[k] = {"tensors_that_differ": len(full), "largest": dict(sorted(full.items(), key=lambda kv: -kv[1])[:8])}
print(json.dumps(res))
