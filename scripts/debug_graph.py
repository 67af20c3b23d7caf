import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np, torch
from sessd_hip import configs, synth, ops
from sessd_hip.engine import InferenceEngine
mode = sys.argv[1]
dev = torch.device("cuda:0"); VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20000, dev)
frames = [torch.from_numpy(synth.make_frame(i, 20000)).to(dev) for i in range(16)]
print([f.shape[0] for f in frames], flush=True)
eng.set_points([frames[0]]); eng.enqueue(); torch.cuda.synchronize()
eng.capture(); print("captured", flush=True)
if mode == "sync":
    for i in range(20):
        eng.set_points([frames[i % 16]]); eng.replay(); torch.cuda.synchronize()
        print("step", i, "ok", int(eng.out["count"][0].item()), int(eng.prefix[1].item()), [int(L["n"].item()) for L in eng.levels[1:]], flush=True)
elif mode == "nosync":
    for n in (2, 4, 8, 16, 32, 64):
        t = time.time()
        for i in range(n):
            eng.set_points([frames[i % 16]]); eng.replay()
        torch.cuda.synchronize()
        print("burst", n, "ok %.1f ms/step" % ((time.time() - t) / n * 1e3), flush=True)
elif mode == "nosync_same":
    for n in (2, 4, 8, 16, 32, 64):
        t = time.time()
        for i in range(n):
            eng.replay()
        torch.cuda.synchronize()
        print("burst-same", n, "ok %.1f ms/step" % ((time.time() - t) / n * 1e3), flush=True)
