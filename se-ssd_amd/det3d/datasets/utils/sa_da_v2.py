"""mirrors det3d/datasets/utils/sa_da_v2.py: SE-SSD's shape-aware data augmentation. Every ground-truth box is cut into six
pyramids (apex = box centre, base = one face); per box, at random: one pyramid loses its points (dropout), one dense
pyramid is thinned to a fixed count by farthest-point sampling (sparsify), and one pyramid trades its points with the same
pyramid of another box, re-expressed in the receiving pyramid's frame with the intensity range of the points it replaces
(swap). Random draws follow the reference's order on numpy's global generator.

The thinning calls `ifp_sample` of the external package jackd/ifp-sample (README.md:53-54; no version pinned, absent here).
With the all-pairs neighbourhoods the reference hands it (cKDTree.query with k = sample size) it is plain iterative
farthest-point sampling, restated in `ifp_sample` below with the first point as seed: parity of that one choice is unpinned."""
import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit

_BASE = np.array([[0, 1, 5, 4], [4, 5, 6, 7], [7, 6, 2, 3], [3, 2, 1, 0], [1, 2, 6, 5], [0, 4, 7, 3]])
_PYRAMID_FACES = [1, 2, 0, 2, 3, 0, 3, 4, 0, 4, 1, 0, 4, 3, 2]


DEVICE_FPS_MAX_POINTS = 4096  # sessd_farthest_point_sample keeps a pyramid's points and distances in LDS


def one_hot(x, num_class=None):
    if not num_class:
        num_class = np.max(x) + 1
    out = np.zeros((len(x), num_class))
    out[range(len(x)), x] = 1
    return out


def get_pyramids(gt_boxes):
    """(N, 7) boxes -> (N, 6, 15): apex (box centre) followed by the four base corners of each face, face order front, left,
    back, right, top, bottom as drawn in sa_da_v2.py:20-62."""
    corners = box_np_ops.center_to_corner_box3d(gt_boxes[:, 0:3], gt_boxes[:, 3:6], gt_boxes[:, 6], origin=[0.5, 0.5, 0.5], axis=2)
    apex = np.broadcast_to(gt_boxes[:, None, None, 0:3], (gt_boxes.shape[0], 6, 1, 3))
    return np.concatenate([apex, corners[:, _BASE]], axis=2).reshape(gt_boxes.shape[0], 6, 15)


def points_in_pyramids_mask(points, pyramids):
    """(P, >=3) points, (M, 15) pyramids -> (P, M) bool (four side faces and the base, normals inwards; sa_da_v2.py:65-74)."""
    v = pyramids.reshape(-1, 5, 3)
    return points_in_convex_polygon_3d_jit(points[:, :3], v[:, _PYRAMID_FACES].reshape(-1, 5, 3, 3))


def ifp_sample(dists, indices, out_size):
    """Iterative farthest-point sampling over precomputed neighbourhoods: dists / indices (n, k) = each point's k nearest
    neighbours (itself first). Starts at point 0; every pick lowers its neighbours' distance-to-selection, the next pick is
    the point with the largest remaining distance (lowest index on ties)."""
    n = dists.shape[0]
    if out_size > n:
        raise ValueError("cannot sample more points than there are")
    remaining = np.full((n,), np.inf)
    out = np.empty((out_size,), dtype=np.int64)
    for s in range(out_size):
        i = int(np.argmax(remaining))
        out[s] = i
        np.minimum.at(remaining, indices[i], dists[i])
        remaining[i] = -np.inf
    return out


def get_points_ratio(points, pyramid):
    """coordinates of the points in the pyramid's own frame: along two base edges and along the base-centre -> apex axis
    (sa_da_v2.py:208-214)."""
    base_c = (pyramid[3:6] + pyramid[6:9] + pyramid[9:12] + pyramid[12:]) / 4.0
    e0, e1, ax = pyramid[6:9] - pyramid[3:6], pyramid[12:] - pyramid[3:6], pyramid[0:3] - base_c
    rel = points[:, 0:3] - pyramid[3:6]
    return [(rel * e0).sum(-1) / np.power(e0, 2).sum(), (rel * e1).sum(-1) / np.power(e1, 2).sum(),
            ((points[:, 0:3] - base_c) * ax).sum(-1) / np.power(ax, 2).sum()]


def recover_points_by_ratio(points_ratio, pyramid):
    """inverse of get_points_ratio in another pyramid (sa_da_v2.py:216-221)."""
    a, b, g = points_ratio
    base_c = (pyramid[3:6] + pyramid[6:9] + pyramid[9:12] + pyramid[12:]) / 4.0
    e0, e1, ax = pyramid[6:9] - pyramid[3:6], pyramid[12:] - pyramid[3:6], pyramid[0:3] - base_c
    return (a[:, None] * e0 + b[:, None] * e1) + pyramid[3:6] + g[:, None] * ax


def recover_points_intensity_by_ratio(points_intensity_ratio, max_intensity, min_intensity):
    return points_intensity_ratio * (max_intensity - min_intensity) + min_intensity


def _intensity_ratio(p):
    lo, hi = p[:, -1:].min(), p[:, -1:].max()
    return (p[:, -1:] - lo) / np.clip(hi - lo, 1e-6, 1)


def pyramid_augment_v0(gt_boxes, points, enable_sa_dropout=0.1, enable_sa_sparsity=[0.05, 50], enable_sa_swap=[0.05, 50]):
    """(N, 7) boxes, (P, 4) points -> augmented float32 points (sa_da_v2.py:76-205). A box that was chosen for dropout takes
    no part in the later stages, one chosen for thinning none in the swap."""
    from scipy.spatial import cKDTree
    pyramids = get_pyramids(gt_boxes)
    if enable_sa_dropout is not None and gt_boxes.shape[0] > 0:
        which = one_hot(np.random.randint(0, 6, (pyramids.shape[0])), num_class=6)
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= enable_sa_dropout
        sel = (box_sel[:, None] * which) > 0
        points = points[~points_in_pyramids_mask(points, pyramids[sel]).any(-1)]
        pyramids = pyramids[~box_sel]

    if enable_sa_sparsity is not None and pyramids.shape[0] > 0:
        prob, keep_num = enable_sa_sparsity
        which = one_hot(np.random.randint(0, 6, (pyramids.shape[0])), num_class=6)
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= prob
        sel = (box_sel[:, None] * which) > 0
        counts = points_in_pyramids_mask(points, pyramids.reshape(-1, 15)).sum(0)
        sel = sel & (counts > keep_num).reshape(-1, 6)
        chosen = pyramids[sel]
        if chosen.shape[0] > 0:
            m = points_in_pyramids_mask(points, chosen)
            rest, thinned = points[~m.any(-1)], []
            for k in range(m.shape[1]):
                part = points[m[:, k]]
                d, idx = cKDTree(part[:, 0:3]).query(part[:, 0:3], part.shape[0])
                thinned.append(part[ifp_sample(d, idx, keep_num)])
            points = np.concatenate([rest] + thinned, axis=0)
        pyramids = pyramids[~box_sel]

    if enable_sa_swap is not None:
        prob, min_num = enable_sa_swap
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= prob
        if box_sel.sum() > 0:
            counts = points_in_pyramids_mask(points, pyramids.reshape(-1, 15)).sum(0).reshape(pyramids.shape[0], -1)
            dense = counts > min_num                    # pyramids with enough points to be worth swapping
            cand = dense * box_sel[:, None]
            if cand.sum() > 0:
                bi, pj = np.nonzero(cand)
                pick = [np.random.choice(pj[bi == i]) if e and (bi == i).any() else 0 for i, e in enumerate(box_sel)]
                give = cand * one_hot(pick, num_class=6) == 1      # per selected box one dense pyramid
                src = pyramids[give]
                bi, pj = np.nonzero(give)
                dense[give] = False
                partner = np.array([np.random.choice(np.where(dense[:, j])[0]) if np.where(dense[:, j])[0].shape[0] > 0 else bi[i]
                                    for i, j in enumerate(pj.tolist())])
                dst = pyramids[partner.astype(np.int32), pj.astype(np.int32)]   # same face of another box (itself if none)
                m = points_in_pyramids_mask(points, np.concatenate([src, dst], axis=0))
                rest, moved, ns = points[~m.any(-1)], [], dst.shape[0]
                for k in range(ns):
                    a_pts, b_pts = points[m[:, k]], points[m[:, k + ns]]
                    a_int, b_int = _intensity_ratio(a_pts), _intensity_ratio(b_pts)
                    into_a = recover_points_by_ratio(get_points_ratio(b_pts, dst[k]), src[k])
                    into_b = recover_points_by_ratio(get_points_ratio(a_pts, src[k]), dst[k])
                    into_a_i = recover_points_intensity_by_ratio(b_int, a_pts[:, -1:].max(), a_pts[:, -1:].min())
                    into_b_i = recover_points_intensity_by_ratio(a_int, b_pts[:, -1:].max(), b_pts[:, -1:].min())
                    moved += [np.concatenate([into_a, into_a_i], axis=1), np.concatenate([into_b, into_b_i], axis=1)]
                points = np.concatenate([rest] + moved, axis=0)
    return points.astype(np.float32)


def pyramid_augment_v0_device(gt_boxes, points, enable_sa_dropout=0.1, enable_sa_sparsity=[0.05, 50], enable_sa_swap=[0.05, 50]):
    """pyramid_augment_v0 with the cloud on the device: `points` is a (P, C) float32 CUDA tensor and stays there. Every random
    draw is the host function's, in its order and with its shapes (the per-pyramid point counts the decisions need come back as a
    few hundred integers); membership tests run on sessd_points_in_bodies, removals on sessd_points_compact, the thinning on
    sessd_farthest_point_sample, and the pyramid-to-pyramid re-expression of swapped points in float64 on the device like the
    host's numpy arithmetic. Returns the augmented float32 device tensor (rest first, then the thinned / moved groups, as the
    host function orders them)."""
    import torch
    from det3d.core.bbox.geometry import surface_equ_3d_jitv2
    from sessd_hip import ops
    dev = points.device
    points = points.float().contiguous()

    def planes_of(pyr):                               # (M, 15) pyramids -> (M, 5, 4) inward face planes on the device
        v = pyr.reshape(-1, 5, 3)
        surf = v[:, _PYRAMID_FACES].reshape(-1, 5, 3, 3)
        nrm, d = surface_equ_3d_jitv2(surf[:, :, :3, :])
        return torch.from_numpy(np.ascontiguousarray(np.concatenate([nrm, d[..., None]], axis=-1).astype(np.float32))).to(dev)

    def masks_of(pts, pyr):                           # (P, M) bool on the device
        if pyr.shape[0] == 0 or pts.shape[0] == 0:
            return torch.zeros((pts.shape[0], pyr.shape[0]), dtype=torch.bool, device=dev)
        return ops.points_in_bodies(pts, planes_of(pyr))

    def keep_rows(pts, keep):                         # order-preserving compaction
        if pts.shape[0] == 0:
            return pts
        out, n = ops.points_compact(pts, keep)
        return out[: int(n.item())]

    pyramids = get_pyramids(gt_boxes)
    if enable_sa_dropout is not None and gt_boxes.shape[0] > 0:
        which = one_hot(np.random.randint(0, 6, (pyramids.shape[0])), num_class=6)
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= enable_sa_dropout
        sel = (box_sel[:, None] * which) > 0
        if sel.any():
            points = keep_rows(points, ~masks_of(points, pyramids[sel]).any(-1))
        pyramids = pyramids[~box_sel]

    if enable_sa_sparsity is not None and pyramids.shape[0] > 0:
        prob, keep_num = enable_sa_sparsity
        which = one_hot(np.random.randint(0, 6, (pyramids.shape[0])), num_class=6)
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= prob
        sel = (box_sel[:, None] * which) > 0
        all_m = masks_of(points, pyramids.reshape(-1, 15))
        counts = all_m.sum(0).cpu().numpy()
        sel = sel & (counts > keep_num).reshape(-1, 6)
        chosen = pyramids[sel]
        if chosen.shape[0] > 0:
            m = all_m[:, torch.from_numpy(np.nonzero(sel.reshape(-1))[0]).to(dev)]
            rest, thinned = keep_rows(points, ~m.any(-1)), []
            for k in range(m.shape[1]):
                part = keep_rows(points, m[:, k])
                if part.shape[0] <= DEVICE_FPS_MAX_POINTS:
                    thinned.append(part[ops.farthest_point_sample(part, keep_num)])
                else:  # beyond the kernel's LDS-resident limit (a close-range car face can exceed it): the host thinning
                    from scipy.spatial import cKDTree
                    ph = part.cpu().numpy()
                    d, idx = cKDTree(ph[:, 0:3]).query(ph[:, 0:3], ph.shape[0])
                    thinned.append(part[torch.from_numpy(ifp_sample(d, idx, keep_num)).to(dev)])
            points = torch.cat([rest] + thinned, dim=0).contiguous()
        pyramids = pyramids[~box_sel]

    if enable_sa_swap is not None:
        prob, min_num = enable_sa_swap
        box_sel = np.random.uniform(0, 1, (pyramids.shape[0])) <= prob
        if box_sel.sum() > 0:
            counts = masks_of(points, pyramids.reshape(-1, 15)).sum(0).cpu().numpy().reshape(pyramids.shape[0], -1)
            dense = counts > min_num
            cand = dense * box_sel[:, None]
            if cand.sum() > 0:
                bi, pj = np.nonzero(cand)
                pick = [np.random.choice(pj[bi == i]) if e and (bi == i).any() else 0 for i, e in enumerate(box_sel)]
                give = cand * one_hot(pick, num_class=6) == 1
                src = pyramids[give]
                bi, pj = np.nonzero(give)
                dense[give] = False
                partner = np.array([np.random.choice(np.where(dense[:, j])[0]) if np.where(dense[:, j])[0].shape[0] > 0 else bi[i]
                                    for i, j in enumerate(pj.tolist())])
                dst = pyramids[partner.astype(np.int32), pj.astype(np.int32)]
                m = masks_of(points, np.concatenate([src, dst], axis=0))
                rest, moved, ns = keep_rows(points, ~m.any(-1)), [], dst.shape[0]
                t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev)

                def ratio(p, pyr):                    # get_points_ratio, float64 on the device
                    base_c = (pyr[3:6] + pyr[6:9] + pyr[9:12] + pyr[12:]) / 4.0
                    e0, e1, ax = pyr[6:9] - pyr[3:6], pyr[12:] - pyr[3:6], pyr[0:3] - base_c
                    rel = p[:, 0:3].double() - t64(pyr[3:6])
                    return ((rel * t64(e0)).sum(-1) / float(np.power(e0, 2).sum()), (rel * t64(e1)).sum(-1) / float(np.power(e1, 2).sum()),
                            ((p[:, 0:3].double() - t64(base_c)) * t64(ax)).sum(-1) / float(np.power(ax, 2).sum()))

                def recover(r, pyr):                  # recover_points_by_ratio
                    a, b, g = r
                    base_c = (pyr[3:6] + pyr[6:9] + pyr[9:12] + pyr[12:]) / 4.0
                    e0, e1, ax = pyr[6:9] - pyr[3:6], pyr[12:] - pyr[3:6], pyr[0:3] - base_c
                    return (a[:, None] * t64(e0) + b[:, None] * t64(e1)) + t64(pyr[3:6]) + g[:, None] * t64(ax)

                def iratio(p):                        # _intensity_ratio: float32 like the host's numpy on the float32 column
                    lo, hi = p[:, -1:].min(), p[:, -1:].max()
                    return (p[:, -1:] - lo) / torch.clamp(hi - lo, 1e-6, 1)

                for k in range(ns):
                    a_pts, b_pts = keep_rows(points, m[:, k]), keep_rows(points, m[:, k + ns])
                    a_int, b_int = iratio(a_pts), iratio(b_pts)
                    into_a = recover(ratio(b_pts, dst[k]), src[k])
                    into_b = recover(ratio(a_pts, src[k]), dst[k])
                    into_a_i = b_int * (a_pts[:, -1:].max() - a_pts[:, -1:].min()) + a_pts[:, -1:].min()
                    into_b_i = a_int * (b_pts[:, -1:].max() - b_pts[:, -1:].min()) + b_pts[:, -1:].min()
                    moved += [torch.cat([into_a, into_a_i.double()], dim=1), torch.cat([into_b, into_b_i.double()], dim=1)]
                points = torch.cat([rest.double()] + moved, dim=0)
    return points.float().contiguous()
