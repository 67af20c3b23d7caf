"""Engine time with the level-0 site renumbering / the deep operand ring off and on (MI355X)."""
import sys, torch
sys.path.insert(0, "se-ssd_amd"); sys.path.insert(0, ".")
from sessd_hip import configs, synth
from sessd_hip.engine import InferenceEngine
dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0, max_voxels=16000, num_points=20000)
frames = [torch.from_numpy(synth.make_frame(i, 20000)).to(dev) for i in range(8)]
for flag in (False, True):
    e = InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], 16000, configs.TEST_CFG, batch_size=1,
                        max_points_per_frame=20000, device=dev, sort_sites=flag)
    e.set_points([frames[0]]); e.enqueue(); torch.cuda.synchronize(); e.autotune()
    st = e.stage_times(reps=20)
    e.capture()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(20):
        e.set_points([frames[i % 8]]); e.replay()
    t0.record()
    for i in range(200):
        e.set_points([frames[i % 8]]); e.replay()
    t1.record(); torch.cuda.synchronize()
    print("sort_sites=%s  graph %.1f us/frame  eager stages %s" % (flag, t0.elapsed_time(t1) / 200 * 1e3, {k: round(v, 3) for k, v in st.items()}))
