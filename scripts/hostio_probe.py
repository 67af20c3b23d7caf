"""What bounds host-fed inference (pinned host points in, detections on the device ring)? Variants of the per-frame submission on
the bench's default configuration (four engines on two CU-masked halves) and on two plain streams. One MI355X.
  resident   : no H2D at all (the timed region of bench.py)
  copystream : H2D on ONE copy stream, the engine's stream waits for the copy's event (HostFedPipeline)
  copystreams: one copy stream per engine
  instream   : H2D enqueued on the engine's own stream, in front of the frame (no cross-stream event)
  ahead      : like copystream, but the copy of frame i + E is issued right after frame i was submitted (one round ahead)"""
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
resident = [torch.from_numpy(f).to(dev) for f in frames_np]


def build(masked, n):
    engines, streams = [], []
    for k in range(n):
        if masked:
            st, ncu = ops.cu_masked_stream(k % 2, 2, dev)
        else:
            st, ncu = torch.cuda.Stream(), 0
        e = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
        e.cu_budget = ncu
        e.set_points([resident[0]])
        if k == 0:
            with torch.cuda.stream(st):
                e.enqueue(); torch.cuda.synchronize(); e.autotune()
                if masked:
                    e.set_list_shares("whole")
        else:
            e.adopt_tuning(engines[0])
        e.attach_records(8192)
        with torch.cuda.stream(st):
            e.capture()
        engines.append(e); streams.append(st)
    torch.cuda.synchronize()
    return engines, streams


def run(engines, streams, mode, N=600):
    E = len(engines)
    R = 6
    stage = [[torch.empty((20480, 4), dtype=torch.float32, device=dev) for _ in range(R)] for _ in engines]
    consumed = [[None] * R for _ in engines]
    copy_streams = [torch.cuda.Stream() for _ in range(E if mode == "copystreams" else 1)]
    for e in engines:
        e.record_cursor.zero_()
    torch.cuda.synchronize()
    pending = {}

    def issue_copy(i):
        ei, k = i % E, (i // E) % R
        cs = copy_streams[ei % len(copy_streams)]
        src = pinned[i % 8]
        dst = stage[ei][k][:src.shape[0]]
        with torch.cuda.stream(cs):
            if consumed[ei][k] is not None:
                cs.wait_event(consumed[ei][k])
            dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(cs)
        pending[i] = (dst, ev)

    t0 = time.perf_counter()
    if mode == "ahead":
        for i in range(E):
            issue_copy(i)
    for i in range(N):
        ei, k = i % E, (i // E) % R
        e, st = engines[ei], streams[ei]
        if mode == "resident":
            with torch.cuda.stream(st):
                e.set_points([resident[i % 8]]); e.replay()
            continue
        if mode == "instream":
            src = pinned[i % 8]
            dst = stage[ei][k][:src.shape[0]]
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)
                e.set_points([dst]); e.replay()
            continue
        if mode in ("copystream", "copystreams"):
            issue_copy(i)
        dst, ev = pending.pop(i)
        with torch.cuda.stream(st):
            st.wait_event(ev)
            e.set_points([dst])
            done = torch.cuda.Event(); done.record(st)
            consumed[ei][k] = done
            e.replay()
        if mode == "ahead" and i + E < N:
            issue_copy(i + E)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_per_s": round(N / dt, 1), "host_enqueue_ms_per_frame": round(t_enq / N * 1e3, 4)}


out = {}
for cfg, masked, n in (("four_engines_two_cu_halves", True, 4), ("two_plain_streams", False, 2)):
    engines, streams = build(masked, n)
    out[cfg] = {}
    for mode in ("resident", "copystream", "copystreams", "instream", "ahead", "resident"):
        r = run(engines, streams, mode)
        out[cfg].setdefault(mode, []).append(r)
    del engines, streams
print(json.dumps(out, indent=1))
