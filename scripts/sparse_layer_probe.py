"""One sparse conv layer of SpMiddleFHD on the sites / rulebook of a real synthetic frame, launched N times with a fixed tuning
(for rocprofv3 --pmc / --kernel-trace):   python scripts/sparse_layer_probe.py [--stress] --layer 6 --split 2 --depth 3 --reps 20
Prints the HIP-event time per launch, rulebook pairs, useful and (under --pmc) executed FLOPs can be derived from the counters."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    sys.path.insert(0, p)
import torch
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
ap = argparse.ArgumentParser()
ap.add_argument("--stress", action="store_true")
ap.add_argument("--layer", type=int, nargs="+", default=[6])
ap.add_argument("--split", type=int, default=0)
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
pts, mv, B, ss = (200000, 64000, 8, 3) if a.stress else (20000, 16000, 1, 1)
model = configs.build_synthetic_detector(dev, seed=0, max_voxels=mv, num_points=pts, supersample=ss)
frames = [torch.from_numpy(synth.make_frame(i, pts, supersample=ss)).to(dev) for i in range(B)]
e = InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], mv, configs.TEST_CFG, batch_size=B,
                    max_points_per_frame=pts, device=dev)
e.set_points(frames)
e._tuning_sparse = []
e.enqueue()
torch.cuda.synchronize()
todo, e._tuning_sparse = e._tuning_sparse, None
st = torch.cuda.current_stream().cuda_stream
ns = [int(e.prefix[B].item())] + [int(L["n"].item()) for L in e.levels[1:]]
print("sites per level", ns)
for idx, lay, in_feat, nbr, tm, out_li, out_feat in todo:
    if idx not in a.layer:
        continue
    n = ns[out_li]
    kv = lay["ks"][0] * lay["ks"][1] * lay["ks"][2]
    pairs = int((nbr[:kv, :n] >= 0).sum().item())
    tmask = tm[:(n + 15) // 16].cpu().numpy().astype("uint32")
    steps = int(sum(bin(int(v)).count("1") for v in tmask))
    e.sparse_split[idx] = a.split + 256 * a.depth
    for _ in range(3):
        e._sconv(lay, in_feat, nbr, tm, out_li, out_feat, st, idx=idx)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        e._sconv(lay, in_feat, nbr, tm, out_li, out_feat, st, idx=idx)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    useful = 2.0 * pairs * lay["cin"] * lay["cout"]
    executed = 2.0 * steps * 16 * lay["cin"] * lay["cout"]
    print("layer %d %s %d->%d out sites %d pairs %d tile-steps %d | %.1f us | useful %.2f GFLOP = %.1f TF/s | executed (16-row tiles) "
          "%.2f GFLOP = %.1f TF/s = %.1f%% of 157.3 | useful rows %.1f%%"
          % (idx, lay["kind"], lay["cin"], lay["cout"], n, pairs, steps, us, useful / 1e9, useful / us / 1e6, executed / 1e9,
             executed / us / 1e6, executed / us / 1e6 / 157.3 * 100, 100.0 * pairs / max(1, steps * 16)))
