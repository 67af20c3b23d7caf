"""CPU probe for DESIGN.md section 9 item 1: how full are the 16-site output tiles of the sparse-conv kernel, and how much would
a different site order help?  For every layer geometry of SpMiddleFHD on the synthetic 20 k-point frame it counts
    useful    = rulebook pairs (in row, out row, offset)
    executed  = 16 x sum over output tiles of the offsets present in the tile   (what sparse_conv_kernel issues on the MFMA)
    rows/tile = distinct input rows a tile gathers
for several orders of the OUTPUT sites (the input order only moves addresses). Uses the oracle voxelizer and rulebook.
    python scripts/tile_occupancy_probe.py [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
from oracle import capi, sparse_conv as osc  # noqa: E402
from sessd_hip import synth  # noqa: E402


def morton2(y, x):
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 8)) & np.uint64(0x00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x33333333)
        v = (v | (v << 1)) & np.uint64(0x55555555)
        return v
    return (spread(y) << np.uint64(1)) | spread(x)


def orders(out_idx, first_touch):
    z, y, x = out_idx[:, 1].astype(np.int64), out_idx[:, 2].astype(np.int64), out_idx[:, 3].astype(np.int64)
    o = {"device (first touch)": first_touch,
         "linear z,y,x": np.lexsort((x, y, z)),
         "grid row z,y (x shuffled)": np.lexsort((np.random.RandomState(0).permutation(len(x)), y, z)),   # what site_renumber.hip produces
         "linear y,x,z": np.lexsort((z, x, y)),
         "morton(y,x) then z": np.lexsort((z, morton2(y, x))),
         "4x4 xy patch, z inside": np.lexsort((x % 4, y % 4, z, x // 4, y // 4)),
         "2x8 xy patch, z inside": np.lexsort((x % 8, y % 2, z, x // 8, y // 2))}
    return o


def mask_orders(pairs, out_idx):
    """spconv-v2 style: group sites with the same set of present offsets. Plain mask sort destroys locality (distinct gathered
    rows per tile double); sorting by mask inside blocks of linearly ordered sites keeps the working set of a block in L2."""
    n = len(out_idx)
    mask = np.zeros(n, np.int64)
    for k, (ri, ro) in enumerate(pairs):
        mask[ro] |= (1 << k)
    z, y, x = out_idx[:, 1].astype(np.int64), out_idx[:, 2].astype(np.int64), out_idx[:, 3].astype(np.int64)
    lin = np.lexsort((x, y, z))
    o = {"offset-mask sort (global)": np.lexsort((x, y, z, mask))}
    for blk in (256, 1024):
        o["linear, mask-sorted in %d-blocks" % blk] = lin[np.lexsort((mask[lin], np.arange(n) // blk))]
    return o


def stats(pairs, n_out, order):
    rank = np.empty(n_out, np.int64)
    rank[order] = np.arange(n_out)
    tiles = (n_out + 15) // 16
    useful, executed, rows = 0, 0, []
    tile_rows = [[] for _ in range(0)]
    seen_rows = {}
    present = np.zeros((tiles, len(pairs)), bool)
    all_t, all_r = [], []
    for k, (ri, ro) in enumerate(pairs):
        t = rank[ro] // 16
        present[t, k] = True
        useful += len(ri)
        all_t.append(t); all_r.append(ri)
    executed = 16 * int(present.sum())
    key = np.unique(np.concatenate(all_t) * (1 << 32) + np.concatenate(all_r))
    distinct = np.bincount((key >> 32).astype(np.int64), minlength=tiles)
    return useful, executed, float(distinct.mean()), float(present.sum(1).mean())


def propagate(idx0, label):
    """Second question: if ONLY level 0 is renumbered (one sort per frame) and every strided conv keeps numbering its output
    sites by first touch, how much of the gain reaches the deeper levels?"""
    idx, shape, tot = idx0, [41, 1600, 1408], [0, 0]
    done = set()
    for (kind, cin, cout, ks, st, pd, key) in osc.SPMIDDLE_FHD_LAYERS:
        if kind == "subm":
            if key in done:
                continue
            done.add(key)
            out_idx, oshape, pairs = osc.rulebook(idx, shape, ks, 1, 0, True)
            order, reps = np.arange(len(idx)), sum(1 for l in osc.SPMIDDLE_FHD_LAYERS if l[6] == key)
        else:
            out_idx, oshape, pairs = osc.rulebook(idx, shape, ks, st, pd, False)
            creator = np.full(len(out_idx), np.iinfo(np.int64).max)
            for ri, ro in pairs:
                np.minimum.at(creator, ro, ri)
            order, reps = np.argsort(creator, kind="stable"), 1
        u, e, rows, offs = stats(pairs, len(out_idx), order)
        tot[0] += u * reps * cin * cout; tot[1] += e * reps * cin * cout
        if kind != "subm":
            # renumber: site r of the next level = out_idx[order[r]]; the rulebook is recomputed from the new table
            idx, shape = out_idx[order].astype(np.int32), oshape
    print("   level 0 in %-28s -> first touch below: useful / executed = %5.1f %%" % (label, 100.0 * tot[0] / tot[1]))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    pts = synth.make_frame(seed, 20000)
    voxels, coors, num = capi.points_to_voxel(pts, [0.05, 0.05, 0.1], [0, -40.0, -3.0, 70.4, 40.0, 1.0], 5, 20000)
    idx = np.concatenate([np.zeros((len(coors), 1), np.int32), coors], 1)
    shape = [41, 1600, 1408]
    idx_level0 = idx.copy()
    done = set()
    tot = {}
    for (kind, cin, cout, ks, st, pd, key) in osc.SPMIDDLE_FHD_LAYERS:
        if kind == "subm":
            if key in done:
                continue
            done.add(key)
            out_idx, oshape, pairs = osc.rulebook(idx, shape, ks, 1, 0, True)
            first = np.arange(len(idx))          # outputs = inputs, in the order the level was created
            name, reps = "%s %dx%d (x%d layers)" % (key, cin, cout, sum(1 for l in osc.SPMIDDLE_FHD_LAYERS if l[6] == key)), sum(1 for l in osc.SPMIDDLE_FHD_LAYERS if l[6] == key)
            nxt = None
        else:
            out_idx, oshape, pairs = osc.rulebook(idx, shape, ks, st, pd, False)
            creator = np.full(len(out_idx), np.iinfo(np.int64).max)
            for ri, ro in pairs:
                np.minimum.at(creator, ro, ri)
            first = np.argsort(creator, kind="stable")   # the device numbers output sites by their first creating input row
            name, reps = "conv %s s%s %d->%d" % (ks, st, cin, cout), 1
            nxt = (out_idx[first], oshape)
        print("\n%s: %d -> %d sites, %d pairs" % (name, len(idx), len(out_idx), sum(len(p[0]) for p in pairs)))
        cand = orders(out_idx, first)
        if kind == "subm":
            cand.update(mask_orders(pairs, out_idx))
        for oname, order in cand.items():
            u, e, rows, offs = stats(pairs, len(out_idx), order)
            print("   %-34s occupancy %5.1f %%   offsets/tile %5.1f   distinct input rows/tile %6.1f" % (oname, 100.0 * u / e, offs, rows))
            t = tot.setdefault(oname, [0, 0])
            t[0] += u * reps * cin * cout; t[1] += e * reps * cin * cout
        if nxt is not None:
            idx, shape = nxt[0].astype(np.int32), nxt[1]
    print("\nFLOP-weighted over the 14 layers:")
    for oname, (u, e) in tot.items():
        print("   %-34s useful / executed MFMA rows = %5.1f %%" % (oname, 100.0 * u / e))
    print("\nRenumbering level 0 only:")
    for oname, order in orders(idx_level0, np.arange(len(idx_level0))).items():
        propagate(idx_level0[order], oname)


if __name__ == "__main__":
    main()
