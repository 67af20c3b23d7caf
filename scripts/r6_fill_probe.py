"""Round 6: what the fill launch of a frame spends on each of its jobs (one job per launch, HIP events; CU-masked half)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
import torch
from sessd_hip import configs, ops, synth
from sessd_hip.engine import InferenceEngine

dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
st, ncu = ops.cu_masked_stream(0, 2, dev)
e = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
e.cu_budget = ncu
torch.cuda.set_stream(st)
e.set_points([torch.from_numpy(synth.make_frame(0, 20000)).to(dev)])
e.force_active_tiles()
e.set_list_shares("whole")
e.enqueue()
torch.cuda.synchronize()
act = e._active_layers()
fj = e._fill_jobs(act)
names = []
for l in act:
    if l in (e.ACTIVE_PAIR, e.ACTIVE_CONV):
        names += [e.ACTIVE_SLOTS[l][0] + ".a", e.ACTIVE_SLOTS[l][0] + ".b"]
    elif not (e.coarse_fill and l == 5 and 7 in act):
        names.append(e.ACTIVE_SLOTS[l][0])
def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = {"all_jobs_us": timed(lambda: e.ta.fill(fj[0], fj[1], layers=fj[2], tiles=fj[3], near=fj[4], near_kind=fj[5]))}
L4 = e.levels[-1]
out["activity_us"] = timed(lambda: e.ta.run(L4["indices"], L4["n"], L4["cap"]))
per = {}
for i, nm in enumerate(names):
    sl = slice(i, i + 1)
    e.ta._jobs = None
    per[nm] = {"us": round(timed(lambda: e.ta.fill(fj[0][sl], fj[1][sl], layers=fj[2][sl], tiles=fj[3][sl], near=fj[4][sl], near_kind=fj[5][sl])), 1),
               "map_mb": round(fj[0][i].numel() * 4 / 1e6, 1), "tile": fj[3][i], "near": fj[4][i], "near_kind": fj[5][i]}
out["per_job"] = per
out["fractions"] = {k: round(v, 3) for k, v in e.active_tile_fractions().items()}
print(json.dumps(out, indent=1))
