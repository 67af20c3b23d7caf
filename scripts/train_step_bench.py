"""Times one SE-SSD training iteration of the slice that exists: teacher forward (no grad) + student
forward + backward + flat all-reduce (no-op on 1 GPU) + fused clip/Adam/EMA, batch 4 (BASELINE configs[2]), synthetic
20k-point frames, a stand-in loss on the head outputs (MultiGroupHead.loss is not part of the slice). Prints one JSON
line; with --cpu also times the same iteration through the CPU oracle (torch autograd + oracle/optim.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np
import torch

from sessd_hip import configs, ops, synth, train as strain

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--graph", action="store_true", help="also capture the iteration as ONE hipGraph (capacity-form inputs) and time replays")
ap.add_argument("--replays-only", type=int, default=0, help="profiling target: ONE eager warm-up iteration, the capture, then this many replays and nothing else")
ap.add_argument("--real-loss", action="store_true",
                help="BASELINE configs[2]: labelled batch (sessd_hip.trainbench.labelled_batch), the reference loss as the capacity-form "
                     "device op (MultiGroupHead.loss + consistency loss through sessd_head_loss) inside the captured iteration")
args = ap.parse_args()
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR


def loss_fn(ex, s, t, w):
    p, q = s[0], t[0]
    M = ops.mean_all   # not torch's .mean(): its semaphore memset breaks on graph replay on this stack
    return (M(p["box_preds"].pow(2)) + M(torch.sigmoid(p["cls_preds"])) + 0.2 * M(p["dir_cls_preds"].pow(2)) + M(p["iou_preds"].abs())
            + w * M((p["cls_preds"] - q["cls_preds"]).pow(2)))


if args.real_loss:
    from sessd_hip import trainbench
    if args.replays_only:
        model = configs.build_synthetic_detector(dev, seed=0)
        step = strain.TrainStep(model, None, total_steps=1000)
        ex, cap_ex = trainbench.labelled_batch(dev, args.batch)
        step.capture(cap_ex, warmup=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.replays_only):
            step.replay()
        torch.cuda.synchronize()
        print(json.dumps({"what": "captured SE-SSD training iteration with the reference loss (sessd_head_loss), replays only",
                          "replays": args.replays_only, "graph_ms_per_iter": (time.perf_counter() - t0) / args.replays_only * 1e3,
                          "graph_overflow_flag": int(step.student.backbone.last_err.item()),
                          "loss_overflow_flags": int(step.last_record[ops.HEAD_LOSS_RECORD["overflow"]])}))
        sys.exit(0)
    real, step_r = trainbench.measure(dev, args.batch, steps=args.steps, real_loss=True)
    # the same batch eagerly (capacity form, device schedule) and with the round-3 stand-in loss, for comparison
    ex, cap_ex = trainbench.labelled_batch(dev, args.batch)
    for _ in range(3):
        step_r(cap_ex, device_schedule=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_r(cap_ex, device_schedule=True)
    torch.cuda.synchronize()
    real["capacity_form_eager_ms_per_iter"] = (time.perf_counter() - t0) / args.steps * 1e3
    del step_r
    stand, _ = trainbench.measure(dev, args.batch, steps=args.steps, real_loss=False, standin_loss_fn=loss_fn)
    real["standin_loss_graph_ms_per_iter"] = stand["ms_per_iter"]
    print(json.dumps(real))
    sys.exit(0)
model = configs.build_synthetic_detector(dev, seed=0)
step = strain.TrainStep(model, loss_fn, total_steps=1000)
frames = [synth.make_frame(50 + i, 20000) for i in range(args.batch)]
r = ops.voxelize_batch([torch.from_numpy(f).to(dev) for f in frames], VG["voxel_size"], VG["range"], 5, 16000)
m = int(r["prefix"][args.batch].item())
ex = dict(voxels=r["voxels"][:m], coordinates=r["coors"][:m], num_points=r["num_points"][:m],
          num_voxels=torch.tensor(np.diff(r["prefix"].cpu().numpy())), shape=[[1408, 1600, 40]] * args.batch)
if args.replays_only:
    cap_ex = strain.capacity_example(ex, (int(m * 1.08) + 4095) // 4096 * 4096)
    step.capture(cap_ex, warmup=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.replays_only):
        step.replay()
    torch.cuda.synchronize()
    print(json.dumps({"what": "captured SE-SSD training iteration, replays only", "replays": args.replays_only,
                      "graph_ms_per_iter": (time.perf_counter() - t0) / args.replays_only * 1e3,
                      "graph_overflow_flag": int(step.student.backbone.last_err.item())}))
    sys.exit(0)
for _ in range(3):
    step(ex)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step(ex)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
# the fused update alone (HBM-bound: 36 B per parameter + 4 B for the norm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    step.opt.step(1e-3, 0.9, 5)
e1.record()
torch.cuda.synchronize()
upd_ms = e0.elapsed_time(e1) / 20
n = step.flat_s.numel
out = {"what": "SE-SSD training iteration (slice): teacher fwd + student fwd/bwd + fused update", "batch": args.batch,
       "voxels": m, "ms_per_iter": ms, "samples_per_s": args.batch / ms * 1e3, "params": n,
       "fused_update_ms": upd_ms, "fused_update_GBps": n * 40 / (upd_ms * 1e-3) / 1e9, "fused_update_frac_of_8TBps": n * 40 / (upd_ms * 1e-3) / 8e12}
if args.graph:
    # the same iteration as ONE captured graph: capacity-sized inputs, device-side counts and schedule (TrainStep.capture)
    cap_ex = strain.capacity_example(ex, (int(m * 1.08) + 4095) // 4096 * 4096)  # 8 % headroom over this batch's voxels
    eager_cap = []
    for mode in ("eager capacity form", "graph"):
        if mode == "graph":
            step.capture(cap_ex, warmup=1)
        for _ in range(3):
            step.replay() if mode == "graph" else step(cap_ex, device_schedule=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.replay() if mode == "graph" else step(cap_ex, device_schedule=True)
        torch.cuda.synchronize()
        eager_cap.append((time.perf_counter() - t0) / args.steps * 1e3)
    out["capacity_form_eager_ms_per_iter"], out["graph_ms_per_iter"] = eager_cap
    out["graph_samples_per_s"] = args.batch / eager_cap[1] * 1e3
    out["graph_overflow_flag"] = int(step.student.backbone.last_err.item())
if args.cpu:
    from oracle import capi, dense_head, optim as ooptim, sparse_conv as osc
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    feats, coors = [], []
    for b, pts in enumerate(frames):
        v, c, nn_ = capi.points_to_voxel(pts, VG["voxel_size"], VG["range"], 5, 16000)
        feats.append(capi.vfe_mean(v, nn_, 4))
        coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    feats, coors = torch.from_numpy(np.concatenate(feats, 0)), np.concatenate(coors, 0)

    def fwd(ref, grad):
        convs = [ref["backbone.middle_conv.%d.weight" % (3 * i)] for i in range(14)]
        bns = [{k: ref["backbone.middle_conv.%d.%s" % (3 * i + 1, k)] for k in ("weight", "bias", "running_mean", "running_var")} for i in range(14)]
        with torch.set_grad_enabled(grad):
            bev = osc.spmiddle_fhd(feats, coors, args.batch, [1408, 1600, 40], convs, bns, training=True)
            return dense_head.head_forward(dense_head.ssfa_forward(bev, ref, training=True), ref)

    c0 = time.perf_counter()
    ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    tp = fwd(sd, False)
    sp = fwd(ref, True)
    loss_fn(None, [sp], [tp], 1.0).backward()
    keys = [k for k, v in ref.items() if getattr(v, "grad", None) is not None]
    p = np.concatenate([ref[k].detach().numpy().ravel() for k in keys]); g = np.concatenate([ref[k].grad.numpy().ravel() for k in keys])
    ooptim.adam_true_wd_ema_step(p, g, np.zeros_like(p), np.zeros_like(p), p.copy(), 3e-4, 0.01, 0.95, 0.99, 1e-8, 1, 35.0, 0.0)
    out["cpu_port_ms_per_iter"] = (time.perf_counter() - c0) * 1e3
    out["cpu_threads"] = torch.get_num_threads()
print(json.dumps(out))
