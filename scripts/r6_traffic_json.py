"""profiles/r6_wino_traffic.json (the source of bench.py's `roofline.traffic`) FROM the counter tables, not by hand:

    python scripts/r6_traffic_json.py profiles/r6_dense_pmc_cu_half.txt profiles/r6_dense_pmc_whole_chip.txt > profiles/r6_wino_traffic.json

Tables: scripts/pmc_compact.py over three separate rocprofv3 --pmc passes (SQ set, FETCH_SIZE, WRITE_SIZE) of scripts/sparse_probe.py
--fixed (the same launches in every pass). FETCH_SIZE x 2 (the guide's gfx950 correction for 16-byte-per-lane reads), WRITE_SIZE as
reported -- CALIBRATED in the same tables on launches of known bytes: fill_multi_kernel writes 22.3 MB (control words + hash arena
+ the 18.0 MB BEV map) and must read 22.3 with and without the CU mask; the two-layer full-map launch must write >= 36.0 MB."""
import json
import re
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
        rows[name] = dict(n=int(v[0]), us=v[1], wait_any=v[2], wait_inst=v[3], mfma_busy=v[4], mfma=v[5], fetch_mb=v[6], write_mb=v[7])
    return rows


def fractions(path):
    """the probe's `# active_tile_fractions {...}` line (scripts/sparse_probe.py)"""
    import ast
    for line in open(path):
        if line.startswith("# active_tile_fractions"):
            return ast.literal_eval(line.split("active_tile_fractions", 1)[1].strip())
    return {}


def algorithmic_mb(fr):
    """input + output of the computed tiles + packed U per Winograd launch of the frame: 128 -> 128 @200x176 (18.02 MB maps, U 1.05 MB)
    for block 0 and conv_0 / conv_1, 256 -> 256 @100x88 (9.01 MB maps, U 4.19 MB) for b1.1 / b1.2; full maps where a layer has no list"""
    per = {}
    for nm in ("b0.0", "b0.1", "b0.2"):
        per[nm] = fr.get(nm, 1.0) * 36.04 + 1.05
    for nm in ("b1.1", "b1.2"):
        per[nm] = fr.get(nm, 1.0) * 18.02 + 4.19
    per["conv_0"] = per["conv_1"] = fr.get("conv_0+conv_1", 1.0) * 36.04 + 1.05
    return per


half, whole = table(sys.argv[1]), table(sys.argv[2])
out = {"kernel": "conv3x3s1_winograd_sk_kernel, batch 1: the frame's seven 3x3 stride-1 layers (b0.0 b0.1 b0.2 b1.1 b1.2 over tile lists; conv_0 / conv_1 "
                 "over THEIR list since round 6 -- two launches -- or as one full-map launch of two weight sets)",
       "fetch_correction": "x2 (gfx950 rocprofv3 tallies the 128-byte requests of 16-byte-per-lane reads at 64 B; MI355X_MICROARCH.md); WRITE_SIZE as reported",
       "source": "scripts/r6_traffic_json.py over %s and %s" % tuple(sys.argv[1:3])}
for tag, T, path in (("cu_half_configuration", half, sys.argv[1]), ("whole_chip_configuration", whole, sys.argv[2])):
    fr = fractions(path)
    alg = algorithmic_mb(fr)
    wino = [(k, r) for k, r in T.items() if "winograd_sk_kernel" in k and r["fetch_mb"] is not None and r["write_mb"] is not None]
    lists = [(k, r) for k, r in wino if k.endswith("true>")]
    pair = [(k, r) for k, r in wino if k.endswith("false>")]
    frames, lpf = per_frame(T)
    n_l = sum(lpf[k] for k, _ in lists)
    lf = sum(r["fetch_mb"] * lpf[k] for k, r in lists) / n_l
    lw = sum(r["write_mb"] * lpf[k] for k, r in lists) / n_l
    launches = float(sum(lpf[k] for k, _ in wino))
    total_mb = sum((r["fetch_mb"] + r["write_mb"]) * lpf[k] for k, r in wino)
    alg_total = sum(alg.values())
    fm, fi = T.get("fill_multi_kernel", {}), T.get("fill_inactive_tiles_kernel", {})
    cfg = {"active_tile_fractions": fr, "winograd_launches_per_frame": round(launches, 2),
           "list_launch": {"fetch_mb_corrected": round(lf, 2), "write_mb": round(lw, 2), "per_frame": n_l,
                           "kernels": sorted(k for k, _ in lists)},
           "calibration": {"fill_multi_kernel_write_mb": fm.get("write_mb"), "fill_multi_kernel_known_mb": 22.3,
                           "fill_inactive_tiles_write_mb": fi.get("write_mb")},
           "winograd_mb_per_frame": round(total_mb, 2), "algorithmic_mb_per_frame": round(alg_total, 2),
           "average_per_launch_mb": round(total_mb / launches, 2), "algorithmic_mb_per_launch": round(alg_total / launches, 2),
           "times_algorithmic": round(total_mb / alg_total, 2)}
    if pair:
        pk, pr = max(pair, key=lambda kr: kr[1]["n"])
        cfg["full_map_pair_launch"] = {"kernel": pk, "fetch_mb_corrected": pr["fetch_mb"], "write_mb": pr["write_mb"], "us_profiled": pr["us"],
                                       "mfma_per_launch": pr["mfma"], "write_at_least_the_two_output_maps_36_MB": bool(pr["write_mb"] >= 36.0)}
    out[tag] = cfg
h = out["cu_half_configuration"]
ok = abs((h["calibration"]["fill_multi_kernel_write_mb"] or 0) - 22.3) < 1.0 and \
    h.get("full_map_pair_launch", {}).get("write_at_least_the_two_output_maps_36_MB", True)
use = h if ok else out["whole_chip_configuration"]
out["traffic_bytes"] = int(use["average_per_launch_mb"] * 1e6)
out["algorithmic_bytes_per_launch"] = int(use["algorithmic_mb_per_launch"] * 1e6)
out["traffic_configuration"] = "cu_half_configuration (the timed one; WRITE_SIZE calibration holds under the CU mask)" if ok else \
    "whole_chip_configuration (the CU-half table fails the WRITE_SIZE sanity bounds: not quoted)"
print(json.dumps(out, indent=1))
