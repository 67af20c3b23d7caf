"""TEST INFRASTRUCTURE: train on the synthetic scans, then hold the ENGINE to the ORACLE on TRAINED weights.

Round-4 review ("Missing 1 - 2", "Next round 3"): every parity figure so far was on seeded RANDOM weights (boxes decoded to
kilometres, hence the relaxed `synthetic` comparison rule), nothing showed a loss going down, and `north_star`'s "AP within +-0.1
of the reference" had no reachable proxy. This module connects the pieces the repository already owns:

  1. sessd_hip.trainloop.fit: N captured SE-SSD iterations (teacher + student forward, reference loss, backward, clip / Adam /
     EMA) on FRESH labelled batches of the synthetic scans (device data path inside the clock);
  2. checks on the run: the loss's moving average falls, positives / matched teacher-student boxes / consistency loss are > 0
     late in training, the teacher equals EMA(student) carried beside it in plain torch arithmetic;
  3. on held-out scans, with the TRAINED student's state_dict: detections through the HIP engine and through the CPU oracle
     pipeline (oracle/pipeline.py), frame by frame under oracle/compare.py's STRICT rule (absolute sizes, <= 6 listed decisions:
     the rule for real weights), and both detection sets through the KITTI evaluation (KittiDataset.evaluation, f-3 kernels) ->
     car 3D AP@0.7 (11- and 40-point) of each and their difference.

`python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --out profiles/r5_trained_parity.json` is the
full-size record; tests/test_trained_gpu.py runs a small instance. Imports oracle/ (the checker): lives under tests/."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def moving_average(rows, key, k=5):
    v = np.array([r[key] for r in rows], np.float64)
    if len(v) < k:
        return v
    return np.convolve(v, np.ones(k) / k, mode="valid")


def engine_detections(model, frames_np, dev, active=True, autotune=False):
    """every frame through a batch-1 InferenceEngine built from the (eval-mode) model; returns (detections, engine)"""
    from sessd_hip import configs
    from sessd_hip.engine import InferenceEngine
    VG = configs.VOXEL_GENERATOR
    model.eval()
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    eng.set_points([torch.from_numpy(frames_np[0]).to(dev)])
    if autotune:
        eng.enqueue()
        torch.cuda.synchronize()
        eng.autotune()
    elif active:
        eng.force_active_tiles()
    out = []
    for f in frames_np:
        eng.set_points([torch.from_numpy(f).to(dev)])
        eng.enqueue()
        out.append(eng.results()[0])
    return out, eng


def oracle_detections(state, frames_np, threads=None):
    from oracle import pipeline, postprocess as pp
    from sessd_hip import configs
    VG = configs.VOXEL_GENERATOR
    # (left at torch's default the oracle oversubscribes a many-core host: 3.1 s per frame on the GPU box instead of 0.23 s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, threads or 16)))
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    dets, dbg = [], []
    for f in frames_np:
        want, inter = pipeline.run_frames([f], state, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
        dets.append(want[0])
        dbg.append(inter["debug"][0])
    return dets, dbg


def run(dev, iterations=2000, scenes=400, heldout=200, batch=4, seed=0, log_every=50, workers=8, lr_max=3e-3, save=None,
        ema_check=True, verbose=False, autotune_engine=False):
    from oracle.compare import compare_detections
    from sessd_hip import configs, trainloop
    t_all = time.perf_counter()
    train_pool = trainloop.ScenePool(range(1000, 1000 + scenes), 20000, workers=workers)
    val_pool = trainloop.ScenePool(range(50000, 50000 + heldout), 20000, workers=workers)
    t_gen = time.perf_counter() - t_all
    model = configs.build_synthetic_detector(dev, seed=seed)
    before_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    on_log = (lambda r: print("[train] it %5d total %.4f loss %.4f cons %.4f (w %.3f) pos %d matched %d cand %d/%d"
                              % (r["iteration"], r["total"], r["loss"], r["consistency_loss"], r["consistency_weight"], r["num_pos"],
                                 r["matched_boxes"], r["candidates"], r["candidates_ema"]), flush=True)) if verbose else None
    step, rep = trainloop.fit(model, train_pool, iterations, batch, lr_max=lr_max, seed=seed, log_every=log_every, on_log=on_log,
                              ema_check=ema_check)
    log = rep["log"]
    ma = moving_average(log, "loss", 5)
    third = max(1, len(log) // 3)
    late = log[-third:]
    checks = {
        "loss_first_window": float(np.mean([r["loss"] for r in log[:third]])),
        "loss_last_window": float(np.mean([r["loss"] for r in late])),
        "moving_average_nonincreasing_share": float(np.mean(np.diff(ma) <= 0)) if len(ma) > 1 else None,
        "moving_average_first_last": [float(ma[0]), float(ma[-1])] if len(ma) else None,
        "late_positives_min": float(min(r["positives"] for r in late)),   # (the record's `num_pos` is sample 0's alone, as the reference logs it)
        "late_matched_boxes_mean": float(np.mean([r["matched_boxes"] for r in late])),
        "late_consistency_loss_mean": float(np.mean([r["consistency_loss"] for r in late])),
    }
    # ---- the trained student in eval mode: engine against oracle on held-out scans
    student = step.student
    state = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    moved = float(max((state[k].float() - before_state[k].float()).abs().max() for k in state if state[k].dtype.is_floating_point))
    if save:
        os.makedirs(os.path.dirname(os.path.abspath(save)), exist_ok=True)
        torch.save(state, save)
    # release the training graph's memory before the engines come up
    step.graph = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got, eng = engine_detections(student, val_pool.frames, dev, active=True, autotune=autotune_engine)
    t_eng = time.perf_counter() - t0
    t0 = time.perf_counter()
    want, dbg = oracle_detections(state, val_pool.frames)
    t_orc = time.perf_counter() - t0
    cmp = {"frames": len(got), "identical": 0, "flipped_near_threshold": 0, "mismatch": [], "rule_set": "strict"}
    for i, (g, w, d) in enumerate(zip(got, want, dbg)):
        try:
            r = compare_detections(g, w, d, rule="strict")
            cmp["identical" if not r["flipped"] else "flipped_near_threshold"] += 1
        except AssertionError as ex:
            cmp["mismatch"].append({"frame": i, "why": str(ex)[:300]})
    cmp["matched"] = cmp["identical"] + cmp["flipped_near_threshold"]
    cmp["ok"] = cmp["matched"] == cmp["frames"]
    val = trainloop.SyntheticKitti(val_pool)
    ap_eng, ap_orc = val.evaluate(got), val.evaluate(want)
    diff = {k: [abs(a - b) for a, b in zip(ap_eng[k], ap_orc[k])] for k in ("ap3d_11", "ap3d_40", "bev_11", "bev_40")}
    sizes = np.concatenate([np.asarray(g["box3d_lidar"]).reshape(-1, 7)[:, 3:6] for g in got] + [np.zeros((0, 3))])
    out = {
        "what": "SE-SSD trained from seeded random weights on synthetic ray-cast scans (visible cars as ground truth), then the "
                "trained student evaluated on held-out scans through the HIP engine and through the CPU oracle pipeline",
        "recipe": {"seed": seed, "iterations": iterations, "batch": batch, "train_scene_seeds": [1000, 1000 + scenes],
                   "heldout_scene_seeds": [50000, 50000 + heldout], "points_per_scan": 20000, "lr_max": lr_max,
                   "schedule": "OneCycle (div 10, pct 0.4, moms 0.95 / 0.85), Adam true weight decay 0.01, clip 35, EMA teacher; "
                               "consistency weight: sigmoid ramp-up over the first quarter",
                   "augmentation": "global flip / rotation +-pi/4 / scale 0.95..1.05 on the student's cloud (device)",
                   "init": "configs.build_synthetic_detector(seed)", "min_points_per_gt_car": trainloop.MIN_POINTS},
        "training": {k: v for k, v in rep.items() if k != "log"}, "training_checks": checks, "training_log": log,
        "largest_parameter_or_buffer_change": moved,
        "engine_vs_oracle_strict": cmp,
        "ap_engine": {k: v for k, v in ap_eng.items() if k != "table"}, "ap_oracle": {k: v for k, v in ap_orc.items() if k != "table"},
        "ap_abs_difference": diff, "ap_table_engine": ap_eng["table"], "ap_table_oracle": ap_orc["table"],
        "car_3d_ap_0p7_moderate": {"engine_11pt": ap_eng["ap3d_11"][1], "oracle_11pt": ap_orc["ap3d_11"][1],
                                   "engine_40pt": ap_eng["ap3d_40"][1], "oracle_40pt": ap_orc["ap3d_40"][1]},
        "ap_within_0p1": bool(max(max(v) for v in diff.values()) <= 0.1),
        "detected_box_sizes_max_m": [float(v) for v in (sizes.max(0) if len(sizes) else np.zeros(3))],
        "active_tile_layers_of_the_engine": sorted(eng.active_tile_fractions()),
        "seconds": {"scene_generation": t_gen, "training": rep["seconds"], "engine_eval": t_eng, "oracle_eval": t_orc,
                    "total": time.perf_counter() - t_all},
    }
    return out, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=2000)
    ap.add_argument("--scenes", type=int, default=400)
    ap.add_argument("--heldout", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workers", type=int, default=12)
    ap.add_argument("--lr-max", type=float, default=3e-3)
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--save", default=None, help="write the trained student's state_dict here (a blob: never committed)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--autotune-engine", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out, _ = run(dev, a.iterations, a.scenes, a.heldout, a.batch, a.seed, a.log_every, a.workers, a.lr_max, a.save, verbose=True,
                 autotune_engine=a.autotune_engine)
    brief = {k: out[k] for k in ("training_checks", "engine_vs_oracle_strict", "ap_engine", "ap_oracle", "ap_abs_difference", "ap_within_0p1",
                                 "seconds")}
    brief["engine_vs_oracle_strict"] = {k: v for k, v in brief["engine_vs_oracle_strict"].items() if k != "mismatch"}
    brief["samples_per_s_sustained"] = out["training"]["samples_per_s"]
    print(json.dumps(brief, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
