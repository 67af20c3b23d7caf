"""profiles/r6_mfma_flops_per_frame.json from the whole-frame counter tables (scripts/pmc_compact.py over scripts/sparse_probe.py --fixed):

    python scripts/r6_mfma_flops_json.py profiles/r6_frame_pmc_cu_half.txt profiles/r6_frame_pmc_whole_chip.txt > profiles/r6_mfma_flops_per_frame.json

EXECUTED matrix-core FLOPs of one frame = SQ_INSTS_MFMA per launch x launches per frame x FLOPs per instruction (v_mfma_f32_32x32x2_f32 =
4096 in the dense kernels, v_mfma_f32_16x16x4_f32 = 2048 in the sparse convs)."""
import json
import sys


def per_frame(T):
    """launches per frame of every kernel: the tail window of a counter pass starts inside a frame, so kernels early in the frame
    have one launch less than the others -- frames = the largest count among the once-per-frame kernels, per kernel round(n / frames)"""
    once = [T[k]["n"] for k in ("ssfa_fuse_head_kernel<22, 32>", "bev_tile_activity_kernel", "chain_emit_kernel", "fill_multi_kernel") if k in T]
    frames = max(once) if once else 1
    return frames, {k: max(1, int(round(r["n"] / float(frames)))) for k, r in T.items()}


def table(path):
    rows = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel") or not line.strip():
            continue
        name, rest = line[:64].strip(), line[64:].split()
        if len(rest) != 8:
            continue
        v = [None if x == "-" else float(x) for x in rest]
        rows[name] = dict(n=int(v[0]), us=v[1], mfma=v[5])
    return rows


out = {}
for tag, p in (("cu_half", sys.argv[1]), ("whole_chip", sys.argv[2])):
    T = table(p)
    frames, lpf = per_frame(T)
    dense = sum(lpf[k] * r["mfma"] * 4096 for k, r in T.items() if r["mfma"] and not k.startswith("sparse_conv")) / 1e9
    sparse = sum(lpf[k] * r["mfma"] * 2048 for k, r in T.items() if r["mfma"] and k.startswith("sparse_conv")) / 1e9
    us = sum(lpf[k] * r["us"] for k, r in T.items() if r["us"])
    out[tag] = {"frames_in_the_tail": frames, "launches_per_frame": sum(lpf.values()), "dense_stage_executed_gflop": round(dense, 2), "sparse_convs_executed_gflop": round(sparse, 2),
                "executed_gflop_per_frame": round(dense + sparse, 2), "kernel_us_per_frame_profiled": round(us, 1)}
out["what"] = ("EXECUTED matrix-core FLOPs of one frame from SQ_INSTS_MFMA (scripts/r6_mfma_flops_json.py over %s, %s: instructions per launch x "
               "launches per frame x FLOPs per instruction). bench.py's analytic roofline.dense_stage_executed_gflop counts the same work without the "
               "padding of a list's last 32-tile block; roofline.frac_chip_timed_region_from_counters = executed_gflop_per_frame / ms_per_step / 157.3."
               % tuple(sys.argv[1:3]))
print(json.dumps(out, indent=1))
