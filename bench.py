#!/usr/bin/env python
"""Benchmark of the SE-SSD inference hot path on MI355X: voxelize -> SpMiddleFHD -> SSFA -> heads -> rotated NMS.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is ONE frame of BASELINE.json configs[1] ("Single MI355X inference, KITTI car voxel grid [1600,1408,40],
max 16000 voxels, batch=1") through the whole path, input points already resident in HBM, detections left on the
device (<= 100 boxes). Frames shard across ranks with no data-path collective (weak scaling: every rank runs K
frames); value = total frames / max-over-ranks wall time. By default TWO frames are in flight per GPU (two independent
batch-1 engines on two HIP streams, `--streams 1` for strictly sequential frames): a batch-1 layer is 4.3 wave tiles per
SIMD, so the tail of one frame's kernels overlaps the other's. One JSON line on rank 0, with
  roofline      the dominant kernel (fused-Winograd f32-MFMA 3x3 conv 128->128 @200x176, 5 launches per frame + 2 of the
                same FLOPs at 256->256 @100x88) against the dense f32 MFMA peak: ALGORITHMIC (direct-convolution) FLOPs
                / launch time measured live with HIP events on the launching stream; `mfma_executed_*` = the 16/36 of
                them Winograd actually multiplies; `traffic` = rocprofv3 FETCH_SIZE/WRITE_SIZE (profiles/)
  roofline_spmiddle / stages_ms_eager   SURVEY 8(d)'s HBM figure for the sparse stage and per-stage times (informational)
  host_io       PCIe-inclusive latency-mode rate (informational, never `value`)
  cpu_baseline  the CPU oracle pipeline (port of the reference path: the reference itself cannot run here) on a
                bounded sample of the same frames, on this box's host cores.
`--stress` = BASELINE configs[4] (200k points, 64k voxels, batch 8), `--batch B` = B frames per step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
CONV_FLOPS = 2.0 * 200 * 176 * 128 * 128 * 9  # algorithmic FLOPs of one 3x3 128->128 launch at 200x176 (10.38 GFLOP)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--max-voxels", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=1, help="frames per step (BASELINE configs[1] is batch 1)")
    ap.add_argument("--supersample", type=int, default=1, help="ray supersampling of the synthetic scanner (3 for 200k points)")
    ap.add_argument("--stress", action="store_true",
                    help="BASELINE configs[4]: 200k points/frame, max 64000 voxels, batch 8 (a parity/roofline case, not the metric line)")
    ap.add_argument("--pool", type=int, default=16, help="distinct synthetic frames cycled through")
    ap.add_argument("--eager", action="store_true", help="no hipGraph: launch every kernel from Python")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch CPU threads of the baseline (capped by affinity)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--streams", type=int, default=2,
                    help="frames in flight: independent batch-1 engines on separate HIP streams (1 = strictly one frame at a time)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-io", action="store_true", help="skip the informational host-buffer leg (kernel traces of the timed region)")
    ap.add_argument("--sk-workgroups", type=int, default=0,
                    help="persistent workgroups of the stream-K Winograd launches (multiple of 8; 0 = the kernel's default, all CUs). "
                         "224 with two frames in flight leaves 32 CUs to the other stream's small kernels: +2 % frames/s, but the "
                         "kernel then runs 14 % longer per launch -- the default keeps the timed kernel the one the roofline describes")
    ap.add_argument("--no-autotune", action="store_true", help="keep the default conv tilings")
    ap.add_argument("--no-offset-split", action="store_true", help="autotune without the offset-split sparse conv variants")
    ap.add_argument("--no-streamk", action="store_true", help="autotune without the stream-K Winograd variants")
    ap.add_argument("--wino-cfg", type=int, default=0, help="force this tile_cfg (20-23) on the seven 3x3 stride-1 SSFA layers after autotune")
    return ap.parse_args()


def log(*a):
    if os.environ.get("SESSD_BENCH_VERBOSE"):
        print("[bench %.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    args = parse()
    if args.stress:
        args.points, args.max_voxels, args.batch, args.supersample, args.pool = 200000, 64000, 8, 3, 8
        args.streams = 1
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs one process per GPU (torch.distributed.run --nproc-per-node %d)" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from sessd_hip import configs, ops, synth
    from sessd_hip.engine import InferenceEngine
    VG = configs.VOXEL_GENERATOR

    model = configs.build_synthetic_detector(dev, seed=0, max_voxels=args.max_voxels, num_points=args.points, supersample=args.supersample)
    engines = [InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], args.max_voxels,
                               configs.TEST_CFG, batch_size=args.batch, max_points_per_frame=args.points, device=dev)
               for _ in range(max(1, args.streams))]
    streams = [torch.cuda.Stream() for _ in engines] if len(engines) > 1 else [torch.cuda.current_stream()]
    eng = engines[0]
    # frames of this rank, resident in HBM before the clock starts (rank r takes seeds r*pool ...)
    frames_np = [synth.make_frame(rank * args.pool + i, args.points, supersample=args.supersample) for i in range(args.pool)]
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    log("model + engine built")
    def batch_of(i):
        return [frames[(i * args.batch + b) % args.pool] for b in range(args.batch)]

    eng.set_points(batch_of(0))
    eng.enqueue()
    torch.cuda.synchronize()
    first = eng.results()[0]
    log("first frame done:", len(first["scores"]), "detections")
    if not args.no_autotune:
        eng.allow_offset_split = not args.no_offset_split
        eng.allow_streamk = not args.no_streamk
        rep = eng.autotune()
        log("autotuned tile configs:", {k: (v[0], round(v[1], 4)) for k, v in rep.items()})
    if args.wino_cfg:
        for nm in ("b0.0", "b0.1", "b0.2", "conv_0", "conv_1", "b1.1", "b1.2"):
            eng.tile_cfg[nm] = args.wino_cfg
    for e in engines:
        e.sk_workgroups = args.sk_workgroups
    for e in engines[1:]:
        e.tile_cfg = dict(eng.tile_cfg)
        e.sparse_split = dict(eng.sparse_split)
        e.sk_ws = torch.zeros_like(eng.sk_ws) if eng.sk_ws is not None else None  # one stream-K workspace per stream
        e.set_points(batch_of(0))
        e.enqueue()
    torch.cuda.synchronize()
    # every frame leaves a fixed-size detection record on the device; the end-of-job gather of those records (ONE all_gather
    # per tensor over RCCL when N > 1; tools/dist_test.py:150-186) is inside the timed region
    per_engine = (args.warmup + args.steps + len(engines) - 1) // len(engines) * args.batch + args.batch
    for e in engines:
        e.attach_records(per_engine)
    if not args.eager:
        for e, st in zip(engines, streams):
            with torch.cuda.stream(st):
                e.capture()
        torch.cuda.synchronize()
        log("graph captured")

    def step(i):
        e, st = engines[i % len(engines)], streams[i % len(engines)]
        with torch.cuda.stream(st):
            e.set_points(batch_of(i))  # device-to-device staging into the engine's static input buffer
            if args.eager:
                e.enqueue()
            else:
                e.replay()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from sessd_hip import dist as sdist
    for i in range(args.warmup):
        step(i)
    barrier()
    for e in engines:
        e.record_cursor.zero_()
    barrier()
    log("warmup done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    # end-of-job gather of this rank's records (per engine: frames i, i + streams, ...), still inside the timed region
    gathered = []
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            n_e = int(e.record_counts.shape[0])
            gathered.append(sdist.gather_records(e.records, e.record_counts, n_e * world))
    barrier()
    dt = time.perf_counter() - t0
    frames_gathered = sum(min(int(e.record_cursor.item()), int(g[0].shape[1])) for g, e in zip(gathered, engines)) * world
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log("timed region done: %.3f ms/step" % (dt / args.steps * 1e3))
    dets = int(eng.out["count"][0].item())
    if int(eng.err.item()) != 0:
        raise SystemExit("sparse capacity overflow during the benchmark: results invalid")

    out = None
    if rank == 0:
        out = {
            "metric": "KITTI frames/sec (voxelize->backbone->head->NMS)",
            "value": world * args.steps * args.batch / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SE-SSD KITTI-car inference, %d frame(s)/step: %d-point synthetic HDL-64E front-FOV scans, "
                                   "voxel grid [1408,1600,40], max_voxels %d, batch %d (BASELINE.json configs[%d]); "
                                   "seeded random weights, BatchNorm calibrated"
                                   % (args.batch, args.points, args.max_voxels, args.batch, 4 if args.stress else 1),
                       "launch": "eager" if args.eager else "hipGraph replay", "frames_per_rank": args.steps * args.batch,
                       "frames_in_flight": len(engines), "streamk_workgroups": args.sk_workgroups,
                       "parallelism": "frames sharded over %d rank(s), no data-path collective" % world,
                       "detections_last_frame": dets, "detections_first_frame": int(len(first["scores"])),
                       "records_gathered": frames_gathered,
                       "gather": "one all_gather of fixed-size (frames, 100, 9) float32 records + counts per engine at the end "
                                 "of the job, inside the timed region (RCCL when n_gpus > 1; a device-side no-op at n_gpus = 1)"},
        }
        # ---- roofline of the dominant kernel, measured IN THE FRAME: whole frames are enqueued eagerly with a HIP event before
        # and after each dense conv launch on the launching stream (engine.dense_layer_times); avg_launch_ms = mean over the
        # layer's five launches per frame (b0.0, b0.1, b0.2, conv_0, conv_1: 3x3 128->128 @200x176) and 20 frames. A loop over
        # ONE layer on a hot input (round 1) is a best case (66.8 us); this is what the frame pays, and it agrees with the
        # rocprofv3 kernel trace of the timed region under profiles/.
        if not args.no_roofline:
            # the kernel's SEVEN launches of a frame: five 128->128 @200x176 and two 256->256 @100x88 of the same FLOP count --
            # the set the rocprofv3 kernel trace averages under this kernel's name
            names = ("b0.0", "b0.1", "b0.2", "conv_0", "conv_1", "b1.1", "b1.2")
            cfgs = [eng.tile_cfg.get(nm) for nm in names]
            wino = all(c in (20, 21, 22, 23) or (c is None and ops.USE_WINOGRAD) for c in cfgs)
            streamk = sum(1 for c in cfgs if c in (22, 23))
            log("dense tile_cfg:", {k: v for k, v in eng.tile_cfg.items()})
            eng.set_points(batch_of(0))
            lt = eng.dense_layer_times(reps=20)
            # conv_0 + conv_1 may run as ONE launch of twice the work (engine.merge_branch_convs): per-launch figures are averages
            # over the launches that carry the seven layers
            times = [lt[nm] for nm in names if nm in lt] + ([lt["conv_0+conv_1"]] if "conv_0+conv_1" in lt else [])
            nlayers = sum(1 for nm in names if nm in lt) + (2 if "conv_0+conv_1" in lt else 0)
            assert nlayers == len(names)
            kms = sum(times) / len(times)
            flops = CONV_FLOPS * args.batch * nlayers / len(times)
            ach = flops / (kms * 1e-3) / 1e12
            log("roofline kernel: %.3f ms per launch in sequence" % kms)
            kname = (("conv3x3s1_winograd_sk_kernel / conv3x3s1_winograd_kernel (fused Winograd F(2x2,3x3) on f32 MFMA; %d of the 7 "
                      "layers are on the stream-K kernel, as the per-layer autotune chose)" % streamk) if wino
                     else "conv2d_mfma_kernel<9 taps> (direct implicit GEMM on f32 MFMA)")
            exe = ach * (16.0 / 36.0 if wino else 1.0)  # Winograd F(2x2,3x3) multiplies 16 of the 36 products of direct convolution
            out["roofline"] = {"bound": "mfma", "kernel": kname + ": Conv2d 3x3 128->128 @200x176 (5 layers per frame%s) and "
                               "256->256 @100x88 (2 launches, same FLOPs per layer); the seven layers are 72.6 of the frame's 90.8 "
                               "dense GFLOP" % ("; conv_0 and conv_1 as one launch of two weight sets: %d launches" % len(times)
                                                if "conv_0+conv_1" in lt else ""),
                               "achieved": exe, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": exe / F32_MFMA_PEAK_TFLOPS,
                               "frac_definition": "EXECUTED matrix-core FLOPs (what SQ_INSTS_MFMA counts: 16/36 of the direct-"
                                                  "convolution count for the Winograd kernel) / launch time / dense f32 MFMA peak",
                               "avg_launch_ms": kms,
                               "avg_launch_source": "HIP events before / after each of the kernel's %d launches inside 20 whole frames " % len(times) +
                                                    "(eager enqueue; same stream as the kernels; one frame in flight, the same "
                                                    "launch configuration as the timed region unless --sk-workgroups says otherwise)",
                               "dense_launch_ms": {k: round(v, 5) for k, v in lt.items()},
                               "dense_tile_cfg": {k: eng.tile_cfg.get(k) for k in lt},
                               "flops_per_launch_executed": flops * (16.0 / 36.0 if wino else 1.0),
                               "flops_per_launch_algorithmic": flops,
                               "achieved_algorithmic": ach, "frac_algorithmic": ach / F32_MFMA_PEAK_TFLOPS,
                               "frac_algorithmic_note": "direct-convolution FLOPs 2*H*W*Cin*Cout*9 / time: a speed-up figure, not a "
                                                        "utilisation -- it exceeds 1 at batch >= 4",
                               "traffic": None}
            # HBM traffic of that kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), committed
            # under profiles/; it cannot be collected inside this process
            tpath = os.path.join(ROOT, "profiles", ("r2_wino_sk_traffic.json" if streamk else "r1_winograd_traffic.json") if wino
                                 else "r1_conv_traffic.json")
            if os.path.exists(tpath) and args.batch == 1:
                tj = json.load(open(tpath))
                out["roofline"]["traffic"] = tj["traffic_bytes"]
                out["roofline"]["traffic_source"] = tj["source"]
            # ---- per-stage time (eager, events) and the HBM roofline of SpMiddleFHD (algorithmic bytes / time)
            eng.set_points(batch_of(0))
            st = eng.stage_times()
            sp_bytes, sites = eng.spmiddle_algorithmic_bytes()
            out["stages_ms_eager"] = {k: round(v, 4) for k, v in st.items()}
            gbs = sp_bytes / (st["spmiddle"] * 1e-3) / 1e9
            out["roofline_spmiddle"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                        "algorithmic_bytes": sp_bytes, "sites_per_level": sites, "ms": st["spmiddle"],
                                        "mfma": eng.spmiddle_mfma_report(),
                                        "note": "all 14 sparse layers + the site / rulebook chain of one batch, eager launches; at "
                                                "batch 1 the stage is launch/latency-bound, see --stress for the meaningful case. "
                                                "`mfma`: per-layer HIP-event times of the sparse convs alone and their EXECUTED f32 "
                                                "MFMA rate (active 16-site tile x offset steps x 16 x Cin x Cout x 2 FLOP) against the "
                                                "157.3 TFLOP/s peak; counters and HBM traffic: profiles/r2_sparse_pmc_after.txt"}
        # ---- informational: the same frames handed over as HOST numpy buffers and detections read back to the host, one
        # frame at a time (H2D of P*16 B from pinned memory + graph replay + D2H of <= 100 boxes, synchronous per frame).
        # Never part of `value` (inputs are resident in HBM inside the timed region).
        if not args.eager and args.batch == 1 and not args.no_host_io:
            pinned = [torch.from_numpy(f).pin_memory() for f in frames_np[:8]]
            stage = torch.empty((args.points, 4), dtype=torch.float32, device=dev)
            nio = 100
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            for i in range(nio):
                src = pinned[i % len(pinned)]
                dst = stage[:src.shape[0]]
                dst.copy_(src, non_blocking=True)
                eng.set_points([dst])
                eng.replay()
                eng.results()
            out["host_io"] = {"frames_per_s": nio / (time.perf_counter() - h0), "what": "pinned host points -> H2D -> replay -> D2H "
                              "detections, strictly sequential with a host sync per frame (latency mode, 1 frame in flight)"}
        # ---- CPU baseline: the oracle port of the reference path on the host cores of this box (bounded sample)
        if args.cpu_frames > 0 and world == 1:
            from oracle import pipeline, postprocess as pp
            sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            anchors = pp.create_anchors_3d_range().reshape(-1, 7)
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            ncores = max(1, min(avail, args.cpu_threads))
            torch.set_num_threads(ncores)
            budget = time.perf_counter() + args.cpu_seconds
            log("cpu baseline: %d threads of %d visible cores" % (ncores, avail))
            pipeline.run_frames([frames_np[0]], sd, VG["range"], VG["voxel_size"], 5, args.max_voxels, anchors)  # warm-up
            log("cpu warm-up frame done")
            T = {}
            done = 0
            c0 = time.perf_counter()
            while done < args.cpu_frames and (done == 0 or time.perf_counter() < budget):
                pipeline.run_frames([frames_np[done % args.pool]], sd, VG["range"], VG["voxel_size"], 5, args.max_voxels,
                                    anchors, timings=T)
                done += 1
                log("cpu frame", done, T)
            cdt = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": done / cdt, "unit": "frames/s", "cores": ncores, "kind": "port",
                                   "sample": "%d of the same frames through oracle/pipeline.py (C voxelizer + rotated NMS single "
                                             "thread, torch-CPU gather-mm-scatter sparse conv and oneDNN dense convs on %d threads; "
                                             "%d host cores visible)" % (done, ncores, avail),
                                   "stage_ms": {k: v / done * 1e3 for k, v in T.items()}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
