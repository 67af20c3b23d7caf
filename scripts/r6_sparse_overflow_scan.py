"""Round 6: which training batches overflow a sparse level capacity (profiles/r5_trained_parity_seed1.json carried
sparse_overflow_flag = 1)? Every batch of a DeviceBatcher(seed) through the student's voxel reader + SpMiddleFHD in capacity
mode (no training): live sites per level against the capacities, and the error flag."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from sessd_hip import configs, trainloop

def main():
    dev = torch.device("cuda:0")
    scenes, its = int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    pool = trainloop.ScenePool(range(1000, 1000 + scenes), 20000, workers=8)
    model = configs.build_synthetic_detector(dev, seed=0)
    model.train()
    out = {}
    for seed in (0, 1, 2):
        data = trainloop.DeviceBatcher(pool, dev, 4, its, seed=seed)
        worst, bad = None, []
        for it in range(0, its):
            ex = data.load(it)
            for sfx in ("", "_raw"):
                with torch.no_grad():
                    vf = model.reader(ex["voxels" + sfx], ex["num_points" + sfx])
                    model.backbone(vf, ex["coordinates" + sfx], 4, ex["shape"][0], n_dev=ex["num_voxels_dev" + sfx])
                plan = model.backbone._plan
                ns = [int(ex["num_voxels_dev" + sfx].item())] + [int(n.item()) for n in plan.chain.n_dev]
                caps = [int(ex["coordinates" + sfx].shape[0])] + [int(i.shape[0]) for i in plan.chain.indices]
                ratio = [n / c for n, c in zip(ns, caps)]
                e = int(model.backbone.last_err.item())
                if worst is None or max(ratio[1:]) > max(worst["ratio"][1:]):
                    worst = {"it": it, "sfx": sfx, "n": ns, "cap": caps, "ratio": [round(r, 3) for r in ratio], "err": e}
                if e:
                    bad.append({"it": it, "sfx": sfx, "n": ns, "cap": caps, "err": e})
        out[seed] = {"worst": worst, "overflows": len(bad), "first": bad[:5]}
        print(seed, json.dumps(out[seed]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_sparse_overflow_scan.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
