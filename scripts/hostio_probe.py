"""Where does HostFedPipeline.submit spend its host time? One MI355X."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "se-ssd_amd")]
import torch
from sessd_hip import configs, ops, synth
from sessd_hip.engine import InferenceEngine

dev = torch.device("cuda:0")
VG = configs.VOXEL_GENERATOR
model = configs.build_synthetic_detector(dev, seed=0)
frames_np = [synth.make_frame(i, 20000) for i in range(8)]
pinned = [torch.from_numpy(f).pin_memory() for f in frames_np]
engines, streams = [], []
for k in range(2):
    st = torch.cuda.Stream()
    e = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    e.set_points([torch.from_numpy(frames_np[0]).to(dev)])
    e.attach_records(4096)
    with torch.cuda.stream(st):
        e.capture()
    engines.append(e); streams.append(st)
copy_stream = torch.cuda.Stream()
stage = [[torch.empty((20480, 4), dtype=torch.float32, device=dev) for _ in range(4)] for _ in engines]
torch.cuda.synchronize()
T = {"copy": 0.0, "event": 0.0, "wait": 0.0, "set_points": 0.0, "replay": 0.0}
N = 400
t_all = time.perf_counter()
for i in range(N):
    ei = i % 2
    e, st = engines[ei], streams[ei]
    src = pinned[i % 8]
    dst = stage[ei][(i // 2) % 4][:src.shape[0]]
    t0 = time.perf_counter()
    with torch.cuda.stream(copy_stream):
        dst.copy_(src, non_blocking=True)
        t1 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record(copy_stream)
    t2 = time.perf_counter()
    with torch.cuda.stream(st):
        st.wait_event(ev)
        t3 = time.perf_counter()
        e.set_points([dst])
        t4 = time.perf_counter()
        e.replay()
        t5 = time.perf_counter()
    T["copy"] += t1 - t0; T["event"] += t2 - t1; T["wait"] += t3 - t2; T["set_points"] += t4 - t3; T["replay"] += t5 - t4
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print(json.dumps({"frames_per_s": N / dt, "enqueue_ms_per_frame": t_enq / N * 1e3, "host_us_per_frame": {k: v / N * 1e6 for k, v in T.items()}}, indent=1))
