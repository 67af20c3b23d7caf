"""Time the site-chain kernels alone (rocprofv3 --kernel-trace): real level-0 sites of a synthetic frame vs an empty frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    sys.path.insert(0, p)
import torch
from sessd_hip import configs, synth, ops
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
stress = "--stress" in sys.argv
pts, mv, B, ss = (200000, 64000, 8, 3) if stress else (20000, 16000, 1, 1)
steps = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ((3, 1, 1), (2, 1, 1), 0)]
co, n = [], 0
for b in range(B):
    r = ops.voxelize_batch([torch.from_numpy(synth.make_frame(b, pts, supersample=ss)).to(dev)], VG["voxel_size"], VG["range"], 5, mv)
    m = int(r["prefix"][1].item())
    c = r["coors"][:m].clone(); c[:, 0] = b
    co.append(c); n += m
idx = torch.cat(co).contiguous()
cap0 = idx.shape[0]
n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
zero = torch.zeros((1,), dtype=torch.int32, device=dev)
h0 = ops.sparse_hash_build(idx, n_dev, [40, 1600, 1408])
caps = [int(cap0 * 1.5), int(cap0 * 1.5), int(cap0 * 1.2), int(cap0)]
jobs = []
for l in range(4):
    if l < 4: jobs.append((l, l, 3, 1, 1))
    jobs.append((l, l + 1) + tuple(steps[l]))
err = torch.zeros((1,), dtype=torch.int32, device=dev)
ch = ops.SparseChain([41, 1600, 1408], steps, caps, B, jobs, dev)
for nd in (n_dev, zero, n_dev):
    for _ in range(5):
        ch.run(idx, nd.data_ptr(), cap0, h0, err)
    torch.cuda.synchronize()
print("sites", n, [int(c.item()) for c in ch.n_dev], "err", int(err.item()))
