"""profiles/r6_wino_traffic.json (the source of bench.py's `roofline.traffic`) FROM the counter tables, not by hand:

    python scripts/r6_traffic_json.py profiles/r6_dense_pmc_cu_half.txt profiles/r6_dense_pmc_whole_chip.txt > profiles/r6_wino_traffic.json

Tables: scripts/pmc_compact.py over three separate rocprofv3 --pmc passes (SQ set, FETCH_SIZE, WRITE_SIZE) of scripts/sparse_probe.py
--fixed (the same launches in every pass). FETCH_SIZE x 2 (the guide's gfx950 correction for 16-byte-per-lane reads), WRITE_SIZE as
reported -- CALIBRATED in the same tables on launches of known bytes: fill_multi_kernel writes 22.3 MB (control words + hash arena
+ the 18.0 MB BEV map) and must read 22.3 with and without the CU mask; the two-layer full-map launch must write >= 36.0 MB."""
import json
import re
import sys


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


ALG_LIST_MB = 14.0     # a list launch: input + output of the computed tiles + packed U, average of the five
ALG_PAIR_MB = 2 * (18.02 + 18.02) + 2 * 1.05   # conv_0 + conv_1: two 128 x 200 x 176 f32 maps in, two out, two packed U
half, whole = table(sys.argv[1]), table(sys.argv[2])
out = {"kernel": "conv3x3s1_winograd_sk_kernel, batch 1, the frame's six launches: five over tile lists (whole-unit shares) and the full-map launch "
                 "of conv_0 + conv_1 (two weight sets)",
       "fetch_correction": "x2 (gfx950 rocprofv3 tallies the 128-byte requests of 16-byte-per-lane reads at 64 B; MI355X_MICROARCH.md); WRITE_SIZE as reported",
       "source": "scripts/r6_traffic_json.py over %s and %s" % tuple(sys.argv[1:3])}
for tag, T in (("cu_half_configuration", half), ("whole_chip_configuration", whole)):
    lists = [(k, r) for k, r in T.items() if "winograd_sk_kernel" in k and k.endswith("true>") and r["fetch_mb"] is not None and r["write_mb"] is not None]
    pair = [(k, r) for k, r in T.items() if "winograd_sk_kernel" in k and k.endswith("false>") and r["fetch_mb"] is not None and r["write_mb"] is not None]
    n_l = sum(r["n"] for _, r in lists)
    lf = sum(r["fetch_mb"] * r["n"] for _, r in lists) / n_l
    lw = sum(r["write_mb"] * r["n"] for _, r in lists) / n_l
    pk, pr = max(pair, key=lambda kr: kr[1]["n"])
    fm = T.get("fill_multi_kernel", {})
    fi = T.get("fill_inactive_tiles_kernel", {})
    avg = (5 * (lf + lw) + (pr["fetch_mb"] + pr["write_mb"])) / 6.0
    alg = (5 * ALG_LIST_MB + ALG_PAIR_MB) / 6.0
    out[tag] = {"list_launch": {"fetch_mb_corrected": round(lf, 2), "write_mb": round(lw, 2), "times_algorithmic": round((lf + lw) / ALG_LIST_MB, 2),
                                "kernels": [k for k, _ in lists]},
                "full_map_pair_launch": {"kernel": pk, "fetch_mb_corrected": pr["fetch_mb"], "write_mb": pr["write_mb"], "us_profiled": pr["us"],
                                         "mfma_per_launch": pr["mfma"], "times_algorithmic": round((pr["fetch_mb"] + pr["write_mb"]) / ALG_PAIR_MB, 2),
                                         "write_at_least_the_two_output_maps_36_MB": bool(pr["write_mb"] >= 36.0)},
                "calibration": {"fill_multi_kernel_write_mb": fm.get("write_mb"), "fill_multi_kernel_known_mb": 22.3,
                                "fill_inactive_tiles_write_mb": fi.get("write_mb")},
                "average_over_the_six_launches_mb": round(avg, 2), "algorithmic_mb_per_launch": round(alg, 2), "times_algorithmic": round(avg / alg, 2)}
h = out["cu_half_configuration"]
ok = h["full_map_pair_launch"]["write_at_least_the_two_output_maps_36_MB"] and abs((h["calibration"]["fill_multi_kernel_write_mb"] or 0) - 22.3) < 1.0
use = h if ok else out["whole_chip_configuration"]
out["traffic_bytes"] = int(use["average_over_the_six_launches_mb"] * 1e6)
out["algorithmic_bytes_per_launch"] = int(use["algorithmic_mb_per_launch"] * 1e6)
out["traffic_configuration"] = "cu_half_configuration (the timed one; WRITE_SIZE calibration holds under the CU mask)" if ok else \
    "whole_chip_configuration (the CU-half table fails the WRITE_SIZE sanity bounds: not quoted)"
print(json.dumps(out, indent=1))
