import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from sessd_hip import configs, ops, synth, train as strain
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
def _example(seeds, npts, max_voxels):
    frames = [synth.make_frame(s, npts) for s in seeds]
    r = ops.voxelize_batch([torch.from_numpy(f).to(dev) for f in frames], VG["voxel_size"], VG["range"], 5, max_voxels)
    m = int(r["prefix"][len(frames)].item())
    return dict(voxels=r["voxels"][:m], coordinates=r["coors"][:m], num_points=r["num_points"][:m],
                num_voxels=torch.tensor(np.diff(r["prefix"].cpu().numpy())), shape=[[1408, 1600, 40]] * len(frames))
comp = {}
def loss_fn(ex, sp, tp, w):
    p = sp[0]
    M = ops.mean_all if not os.environ.get("DBG_TORCH_MEAN") else torch.mean
    a, b, c, d = M(p["box_preds"].pow(2)), M(torch.sigmoid(p["cls_preds"])), 0.2 * M(p["dir_cls_preds"].pow(2)), M(p["iou_preds"].abs())
    e = 0.1 * w * M((p["cls_preds"] - tp[0]["cls_preds"]).pow(2))
    comp["terms"] = torch.stack([a.detach(), b.detach(), c.detach(), d.detach(), e.detach()])
    return a + b + c + d + e
def make():
    return strain.TrainStep(configs.build_synthetic_detector(dev, seed=0), loss_fn=loss_fn, total_steps=20)
order = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,1,2").split(",")]
seeds = [(61, 62), (63, 64), (65, 66), (67, 68)]
batches = [strain.capacity_example(_example(s, 9000, 8000), 16384) for s in seeds]
print("voxels per batch", [int(b["num_voxels_dev"].item()) for b in batches])
LATE = os.environ.get("DBG_LATE")
THREE = os.environ.get("DBG_THREE")
eager, graph = make(), make()
eager2 = make() if THREE else None
static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batches[order[0]].items()}
graph.capture(static, warmup=1)
gterms = comp["terms"]
eager(batches[order[0]], device_schedule=True)
if eager2: eager2(batches[order[0]], device_schedule=True)
for i in order[1:]:
    for k in ("voxels", "coordinates", "num_points", "num_voxels_dev"):
        static[k].copy_(batches[i][k])
    lg = graph.replay()
    if not LATE:
        torch.cuda.synchronize()
        gt = gterms.cpu().numpy().copy()
    le, _, _ = eager(batches[i], device_schedule=True)
    if eager2: eager2(batches[i], device_schedule=True)
    torch.cuda.synchronize()
    if LATE:
        gt = gterms.cpu().numpy().copy()
        if abs(gt[1]) > 1.0:
            ptr = gterms.data_ptr()
            print("CORRUPT gterms at 0x%x" % ptr, gt)
            snap = torch.cuda.memory_snapshot()
            for seg in snap:
                if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
                    off = seg["address"]
                    for b in seg["blocks"]:
                        if off <= ptr < off + b["size"]:
                            print(" segment pool", seg.get("segment_pool_id"), "size", seg["total_size"], "block size", b["size"], "state", b["state"], "block addr 0x%x" % off)
                        off += b["size"]
            # who else lives nearby: the eager trainers' persistent small tensors
            for name, t in (("eager2.norm_coef", eager2.opt.norm_coef), ("eager.norm_coef", eager.opt.norm_coef), ("eager2.lr_mom", eager2.opt.lr_mom_dev),
                            ("eager2.args", eager2.opt.args_dev), ("graph.norm_coef", graph.opt.norm_coef), ("graph.args", graph.opt.args_dev)):
                print("  %-18s 0x%x %s" % (name, t.data_ptr(), t.cpu().numpy()[:3]))
    et = comp["terms"].cpu().numpy()
    print("batch %d  graph %.6f eager %.6f | graph terms %s | eager terms %s | dparam %.2e err %d" % (
        i, float(lg), float(le), np.round(gt, 5), np.round(et, 5), float((graph.flat_s.data - eager.flat_s.data).abs().max()),
        int(graph.student.backbone.last_err.item())))
