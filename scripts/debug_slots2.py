import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np, torch
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
P, MV, ss = int(sys.argv[1]), int(sys.argv[2]), 3
fa, fb = synth.make_frame(100, P, supersample=ss), synth.make_frame(101, P, supersample=ss)
frames = [fa, fb, fa, fb]
eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=4, max_points_per_frame=P, device=dev)
eng.set_points([torch.from_numpy(f).to(dev) for f in frames])
eng.enqueue()
torch.cuda.synchronize()
pf = eng.prefix.cpu().numpy()
print("prefix", pf, "err", int(eng.err.item()))
a0, a1, c0, c1 = pf[0], pf[1], pf[2], pf[3]
print("coors eq", bool(torch.equal(eng.coors[a0:a1, 1:], eng.coors[c0:c1, 1:])), "vfeat eq", bool(torch.equal(eng.vfeat[a0:a1], eng.vfeat[c0:c1])),
      "nump eq", bool(torch.equal(eng.nump[a0:a1], eng.nump[c0:c1])))
finals = [("feat_a", 16), ("feat_b", 16)], [("feat_a", 32), ("feat_b", 32)], [("feat_a", 64), ("feat_b", 64)], [("feat_a", 64), ("feat_b", 64)]
for li in range(4):
    L = eng.levels[li]
    n = int(pf[4]) if li == 0 else int(L["n"].item())
    idx = L["indices"][:n].cpu().numpy().astype(np.int64)
    shp = L["shape"]
    key = (idx[:, 1] * shp[1] + idx[:, 2]) * shp[2] + idx[:, 3]
    r0 = np.nonzero(idx[:, 0] == 0)[0]; r2 = np.nonzero(idx[:, 0] == 2)[0]
    o0 = r0[np.argsort(key[r0], kind="stable")]; o2 = r2[np.argsort(key[r2], kind="stable")]
    same_sites = len(o0) == len(o2) and np.array_equal(key[o0], key[o2])
    print("level", li, "n", n, "cap", L["cap"], "sites b0", len(o0), "b2", len(o2), "same", same_sites, "dup keys b0", len(o0) - len(np.unique(key[r0])))
    if not same_sites:
        continue
    for name, c in finals[li]:
        f = L[name].view(-1)[:L["cap"] * c].view(L["cap"], c)[:n].cpu().numpy()
        d = np.abs(f[o0] - f[o2])
        bad = np.nonzero(d.max(1) > 0)[0]
        print("   ", name, c, "maxdiff", float(d.max()), "rows differing", len(bad), "of", len(o0), "first rows", o0[bad[:5]], o2[bad[:5]])
bev = eng.bev
print("bev eq", bool(torch.equal(bev[0], bev[2])), float((bev[0] - bev[2]).abs().max()))
