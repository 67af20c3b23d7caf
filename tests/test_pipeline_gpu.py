"""End-to-end: engine (fused, hipGraph-able) and the det3d-mirror module path vs the CPU oracle pipeline
(oracle/pipeline.py) on the same seeded weights and synthetic KITTI-shaped frames.

Tolerances: BEV / SSFA feature maps 2e-4 * max|ref| (float32 sums in another order through 14 + 14 layers);
detections: same count and order, boxes within 2e-3 m / rad, scores within 1e-3 relative. When the oracle reports NMS
decisions within 1e-4 of the 0.01 IoU threshold, the detections must equal the oracle's under SOME assignment of those listed
decisions (at most 10 per frame; oracle/compare.py); no path returns success without comparing every box."""
import numpy as np
import pytest
import torch

from oracle import pipeline, postprocess as pp
from oracle.compare import compare_detections
from sessd_hip import configs, ops, synth
from sessd_hip.engine import InferenceEngine

pytestmark = pytest.mark.gpu
VG = configs.VOXEL_GENERATOR


@pytest.fixture(scope="module")
def model(dev):
    return configs.build_synthetic_detector(dev, seed=0)


@pytest.fixture(scope="module")
def state(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _compare_dets(got, want, dbg):
    """oracle/compare.py: identical detections, or identical to the oracle re-run with some of its LISTED near-threshold NMS
    decisions (|IoU - 0.01| < 1e-4, at most 10) taken the other way. Never returns without having compared every box."""
    r = compare_detections(got, want, dbg, rule="synthetic")  # seeded random weights: the synthetic rule set
    if r["flipped"]:
        print("device == oracle with near-threshold decisions (kept row, candidate row, suppress):", r["flipped"])
    return r


@pytest.mark.parametrize("batch,seeds,max_voxels", [(1, (0,), 16000), (2, (3, 4), 20000)])
def test_engine_vs_oracle(dev, model, state, batch, seeds, max_voxels):
    frames = [synth.make_frame(s, 20000) for s in seeds]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    cal = synth.kitti_calib()
    fr = pp.get_valid_frustum(cal["rect"], cal["Trv2c"], cal["P2"], cal["image_shape"])
    want, inter = pipeline.run_frames(frames, state, VG["range"], VG["voxel_size"], 5, max_voxels, anchors,
                                      [fr] * batch, return_intermediate=True)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, max_voxels, configs.TEST_CFG, batch_size=batch,
                          max_points_per_frame=20480, device=dev, use_frustum=True)
    eng.keep_ssfa = True  # the fused tail + heads launch writes the SSFA output only on request
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames], torch.from_numpy(np.stack([fr] * batch)).to(dev))
    eng.enqueue()
    got = eng.results()
    # intermediate tensors
    n0 = int(eng.prefix[batch].item())
    assert n0 == inter["num_voxels"]
    for li, lvl_idx in ((1, 2), (2, 5), (3, 9), (4, 13)):
        assert int(eng.levels[li]["n"].item()) == inter["levels"][lvl_idx][0].shape[0]
    bev = eng.bev.cpu()
    assert float((bev - inter["bev"]).abs().max()) < 2e-4 * max(1.0, float(inter["bev"].abs().max()))
    ssfa = eng.t["out"].cpu()
    assert float((ssfa - inter["ssfa"]).abs().max()) < 5e-4 * max(1.0, float(inter["ssfa"].abs().max()))
    res = [_compare_dets(g, w, d) for g, w, d in zip(got, want, inter["debug"])]
    assert all(r["matched"] == r["n"] for r in res)
    print("detections per frame", [len(g["scores"]) for g in got], "candidates", [d["num_candidates"] for d in inter["debug"]])


@pytest.mark.parametrize("cfg,wgs", [(22, 0), (23, 0), (22, 224), (30, 0)])
def test_engine_with_stream_k_dense_layers_vs_oracle(dev, model, state, cfg, wgs):
    """The configuration bench.py times: the seven 3x3 stride-1 SSFA layers on the stream-K Winograd kernel (what
    engine.autotune() selects on MI355X), eagerly and through a captured graph replayed twice -- same oracle comparison as the
    default engine, and the replays must reproduce the eager bits (fixed shape and workgroup count => fixed summation order)."""
    frames = [synth.make_frame(s, 20000) for s in (21, 22)]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    want, inter = pipeline.run_frames(frames, state, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, batch_size=2, max_points_per_frame=20480,
                          device=dev)
    if cfg == 30:  # the other dense layers on the LDS-tiled stream-K kernel (the 3x3 stride-1 ones stay on Winograd stream-K)
        for name in ("b1.0", "trans_0", "trans_1", "deconv_0", "deconv_1"):
            eng.tile_cfg[name] = 30
        cfg = 22
    else:  # the two transposed convs as ONE launch (engine.merge_branch_convs; same bits as two launches)
        eng.tile_cfg["deconv_0"] = eng.tile_cfg["deconv_1"] = 4
    for name in ("b0.0", "b0.1", "b0.2", "conv_0", "conv_1", "b1.1", "b1.2"):  # conv_0 + conv_1: one launch of two weight sets
        eng.tile_cfg[name] = cfg
    eng.sk_workgroups = wgs
    eng.sk_ws = torch.zeros(max(ops.winograd_sk_workspace(2, 200, 176, 256, dev, 0, cfg - 22).numel(),
                                ops.conv2d_sk_workspace(2, 200, 176, 256, 4, dev, 0).numel()), dtype=torch.uint8, device=dev)
    eng.keep_ssfa = True  # the fused tail + heads launch writes the SSFA output only on request
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames])
    eng.enqueue()
    got = eng.results()
    ssfa = eng.t["out"].cpu()
    assert float((ssfa - inter["ssfa"]).abs().max()) < 5e-4 * max(1.0, float(inter["ssfa"].abs().max()))
    res = [_compare_dets(g, w, d) for g, w, d in zip(got, want, inter["debug"])]
    assert all(r["matched"] == r["n"] for r in res) and sum(r["n"] for r in res) > 20
    eng.capture()
    for _ in range(2):
        eng.replay()
        again = eng.results()
        for a, b in zip(got, again):
            for k in ("box3d_lidar", "scores", "label_preds"):
                assert np.array_equal(a[k], b[k]), k
    assert int(eng.sk_ws[:4096].view(torch.int32).abs().sum().item()) == 0   # unit counters back at zero


@pytest.mark.parametrize("batch", [1, 2, 8])
def test_whatever_autotune_chooses_equals_the_oracle(dev, model, state, batch):
    """engine.autotune() picks per-layer tilings, sparse variants (incl. the offset split, whose summation order differs in the
    last bit) and stream-K kernels (whose bits depend on shape and workgroup count) BY WALL CLOCK: whatever it chose on this
    device is held to the oracle, eagerly and through graph replay, at batch 1, 2 and 8. Tolerance ACROSS batch sizes of the
    stream-K path (Numerics): features 2e-4 * max|ref| against the oracle; detections identical or identical
    under the oracle's listed near-threshold NMS decisions -- NOT bit-identical between batch sizes."""
    seeds = list(range(40, 40 + batch))
    frames = [synth.make_frame(s, 20000) for s in seeds]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    want, inter = pipeline.run_frames(frames, state, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, batch_size=batch, max_points_per_frame=20480,
                          device=dev)
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames])
    eng.enqueue()
    torch.cuda.synchronize()
    rep = eng.autotune()
    print("autotune chose", {k: v[0] for k, v in rep.items()})
    eng.enqueue()
    got = eng.results()
    bev = eng.bev.cpu()
    assert float((bev - inter["bev"]).abs().max()) < 2e-4 * max(1.0, float(inter["bev"].abs().max()))
    res = [_compare_dets(g, w, d) for g, w, d in zip(got, want, inter["debug"])]
    assert all(r["matched"] == r["n"] for r in res)
    eng.capture()
    eng.replay()
    again = eng.results()
    for a, b in zip(got, again):
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert np.array_equal(a[k], b[k]), k


def test_graph_replay_is_bit_identical_and_idempotent(dev, model):
    frames = [synth.make_frame(7, 20000)]
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    eng.set_points([torch.from_numpy(frames[0]).to(dev)])
    eng.enqueue()
    a = eng.results()[0]
    eng.capture()
    eng.replay()
    b = eng.results()[0]
    eng.replay()
    c = eng.results()[0]
    for k in ("box3d_lidar", "scores"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(b[k], c[k])
    # a different frame through the same captured graph
    eng.set_points([torch.from_numpy(synth.make_frame(8, 18000)).to(dev)])
    eng.replay()
    d = eng.results()[0]
    eng.graph = None
    eng.enqueue()
    e = eng.results()[0]
    assert np.array_equal(d["scores"], e["scores"])


def test_module_path_matches_engine(dev, model):
    """VoxelNet.forward(example, return_loss=False) through the det3d-mirror modules == the fused engine."""
    from sessd_hip import ops
    frame = synth.make_frame(11, 20000)
    pts = torch.from_numpy(frame).to(dev)
    r = ops.voxelize_batch([pts], VG["voxel_size"], VG["range"], 5, 16000)
    m = int(r["prefix"][1].item())
    anchors = torch.from_numpy(pp.create_anchors_3d_range().reshape(1, -1, 7)).to(dev)
    example = dict(voxels=r["voxels"][:m], coordinates=r["coors"][:m], num_points=r["num_points"][:m],
                   num_voxels=torch.tensor([m]), shape=[[1408, 1600, 40]], anchors=[anchors], metadata=[dict(token="0")])
    with torch.no_grad():
        dets = model(example, return_loss=False)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    eng.set_points([pts])
    eng.enqueue()
    ref = eng.results()[0]
    got = dets[0]
    assert got["metadata"] == dict(token="0")
    assert got["box3d_lidar"].shape[0] == ref["box3d_lidar"].shape[0]
    assert np.allclose(got["scores"].cpu().numpy(), ref["scores"], rtol=2e-3, atol=1e-6)
    if ref["box3d_lidar"].shape[0]:
        gb = got["box3d_lidar"].cpu().numpy()
        assert np.allclose(gb, ref["box3d_lidar"], rtol=1e-3, atol=1e-3)


def test_data_pipeline_contract_to_detections(dev, model):
    """Voxelization -> Reformat -> collate_kitti -> example_to_device -> VoxelNet.forward (reference call chain,
    tools/test.py:121-146) on two frames == the fused engine on the same frames."""
    from det3d.datasets.pipelines import Reformat, Voxelization
    from det3d.torchie.parallel import collate_kitti, example_to_device
    from det3d.torchie.utils.config import ConfigDict
    vox = Voxelization(cfg=ConfigDict(range=VG["range"], voxel_size=VG["voxel_size"], max_points_in_voxel=5,
                                      max_voxel_num=16000, far_points_first=False))
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    cal = synth.kitti_calib()
    fr = pp.get_valid_frustum(cal["rect"], cal["Trv2c"], cal["P2"], cal["image_shape"])
    frames = [synth.make_frame(21, 20000), synth.make_frame(22, 17000)]
    examples = []
    for i, f in enumerate(frames):
        res = dict(mode="val", labeled=False, metadata=dict(token=str(i)), calib=dict(frustum=fr),
                   lidar=dict(points=f, targets=dict(anchors=[anchors])))
        res, _ = vox(res, None)
        ex, _ = Reformat()(res, None)
        examples.append(ex)
    batch = example_to_device(collate_kitti(examples), dev)
    assert batch["coordinates"].shape[1] == 4 and int(batch["coordinates"][-1, 0]) == 1
    with torch.no_grad():
        dets = model(batch, return_loss=False)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 2, 20480, dev, use_frustum=True)
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames], torch.from_numpy(np.stack([fr, fr])).to(dev))
    eng.enqueue()
    ref = eng.results()
    for d, r in zip(dets, ref):
        assert d["box3d_lidar"].shape[0] == r["box3d_lidar"].shape[0]
        assert np.allclose(d["scores"].cpu().numpy(), r["scores"], rtol=2e-3, atol=1e-6)
    assert [d["metadata"]["token"] for d in dets] == ["0", "1"]


def test_stress_config_dense_scene(dev):
    """BASELINE.json configs[4] (dense-scene stress): 200k points per frame, max 64000 voxels, batch 8 on one GPU.
    (a) frame 0 against the CPU oracle pipeline at full size; (b) size-independent properties: a frame's detections do
    not depend on its batch slot or on the batch size (bit-identical), voxel counts equal the oracle voxelizer's."""
    from oracle import capi
    B, P, MV = 8, 200000, 64000
    # BatchNorm calibrated on a dense scene (as bench.py --stress does): with the sparse-scan calibration the dense frames decode
    # to boxes hundreds of kilometres long, on which float32 corner geometry -- and so any NMS comparison -- is meaningless
    model = configs.build_synthetic_detector(dev, seed=0, calib_frame_seed=99, max_voxels=MV, num_points=P, supersample=3)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    distinct = [synth.make_frame(100 + i, P, supersample=3) for i in range(3)]
    frames = [distinct[i % 3] for i in range(B)]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=B,
                          max_points_per_frame=P, device=dev)
    eng.set_points([torch.from_numpy(f).to(dev) for f in frames])
    eng.enqueue()
    got = eng.results()  # raises on a sparse-capacity overflow
    prefix = eng.prefix.cpu().numpy()
    for i in range(3):
        v, c, n = capi.points_to_voxel(distinct[i], VG["voxel_size"], VG["range"], 5, MV)
        assert prefix[i + 1] - prefix[i] == c.shape[0] == prefix[i + 4] - prefix[i + 3]
    print("stress: voxels/frame", np.diff(prefix)[:3], "level sites", [int(l["n"].item()) for l in eng.levels[1:]],
          "detections", [len(g["scores"]) for g in got])
    # (b) slot independence inside the batch, and against a batch-1 engine
    for i in range(3, B):
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert np.array_equal(got[i][k], got[i % 3][k]), (i, k)
    eng1 = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=1,
                           max_points_per_frame=P, device=dev)
    eng1.set_points([torch.from_numpy(distinct[1]).to(dev)])
    eng1.enqueue()
    one = eng1.results()[0]
    for k in ("box3d_lidar", "scores", "label_preds"):
        assert np.array_equal(one[k], got[1][k]), k
    # (a) full-size oracle comparison of one frame
    want, inter = pipeline.run_frames([distinct[0]], state, VG["range"], VG["voxel_size"], 5, MV, anchors, None,
                                      return_intermediate=True)
    r = _compare_dets(got[0], want[0], inter["debug"][0])
    assert r["matched"] == r["n"]
    assert float(np.abs(want[0]["box3d_lidar"][:, 3:6]).max(initial=0)) < 50.0  # car-sized boxes, not the degenerate regime


def test_stress_autotuned_equals_the_oracle(dev):
    """BASELINE.json configs[4] in the configuration `bench.py --stress` TIMES: the batch-8 / 200 k-point / 64 k-voxel engine after
    engine.autotune() on that very batch (the dense tilings, stream-K kernels and sparse variants it picks at this size are not
    the batch-1 ones, and round 3 never held them to the oracle), eagerly and through graph replay. Every frame of the batch is
    compared with the CPU oracle at full size (four distinct frames, each in two slots; oracle/compare.py rule), the BEV map of
    every slot to 2e-4 * max|ref|."""
    B, P, MV = 8, 200000, 64000
    model = configs.build_synthetic_detector(dev, seed=0, calib_frame_seed=99, max_voxels=MV, num_points=P, supersample=3)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    distinct = [synth.make_frame(200 + i, P, supersample=3) for i in range(4)]
    order = [0, 1, 2, 3, 2, 0, 3, 1]
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    wants = [pipeline.run_frames([f], state, VG["range"], VG["voxel_size"], 5, MV, anchors, None, return_intermediate=True)
             for f in distinct]
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=B,
                          max_points_per_frame=P, device=dev)
    eng.set_points([torch.from_numpy(distinct[i]).to(dev) for i in order])
    eng.enqueue()
    torch.cuda.synchronize()
    rep = eng.autotune()
    print("stress autotune chose", {k: v[0] for k, v in rep.items()})
    eng.enqueue()
    got = eng.results()
    bev = eng.bev.cpu()
    for slot, i in enumerate(order):
        want, inter = wants[i]
        ref = inter["bev"][0]
        assert float((bev[slot] - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), slot
        r = _compare_dets(got[slot], want[0], inter["debug"][0])
        assert r["matched"] == r["n"], (slot, i)
        assert float(np.abs(want[0]["box3d_lidar"][:, 3:6]).max(initial=0)) < 1000.0  # not the kilometre regime of an uncalibrated model
    assert sum(len(g["scores"]) for g in got) > 20
    eng.capture()
    eng.replay()
    again = eng.results()
    for a, b in zip(got, again):
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert np.array_equal(a[k], b[k]), k


def test_offset_pattern_tiles_do_not_change_the_frame(dev, model):
    """InferenceEngine(sort_tiles=True) (the chain groups the sites of every 256-row group into 16-row tiles by neighbour pattern,
    the sparse convs walk those tiles; measured slower on MI355X and therefore not the default) against sort_tiles=False: the SAME BEV map and detections, bit for bit, at batch 1
    and 2, and a higher share of executed MFMA rows that carry a pair."""
    for B, seeds in ((1, (51,)), (2, (52, 53))):
        frames = [torch.from_numpy(synth.make_frame(s, 20000)).to(dev) for s in seeds]
        res = []
        for srt in (True, False):
            eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, B, 20480, dev, sort_tiles=srt)
            eng.set_points(frames)
            eng.enqueue()
            out = eng.results()
            rep = eng.spmiddle_mfma_report(reps=2)
            res.append((eng.bev.clone(), out, rep["useful_row_fraction"], [r["sorted_tiles"] for r in rep["layers"]]))
        assert torch.equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1], res[1][1]):
            for k in ("box3d_lidar", "scores", "label_preds"):
                assert np.array_equal(a[k], b[k]), k
        assert all(res[0][3]) and not any(res[1][3])
        print("useful MFMA rows: offset-pattern tiles %.3f, plain tiles %.3f" % (res[0][2], res[1][2]))
        assert res[0][2] > res[1][2] + 0.08


def test_active_tiles_do_not_change_the_frame(dev, model, state):
    """Blocks 0 and 1 of the neck in active-tile mode (csrc/dense_active.hip: only the 2x2-output tiles whose input patch is not
    constant are computed, the others are filled with the layer's constant) against the same engine with the whole map computed: the
    blocks' outputs within 2e-6 of their largest value (float32 rounding of the constants' chain; a computed tile is the same
    arithmetic as before), the same detections under the tolerance of the oracle comparison, at batch 1 and 2, eagerly and as a
    captured graph; and the share of computed tiles is what the occupancy of a 20 k-point scan gives (well under half)."""
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    for B, seeds in ((1, (51,)), (2, (52, 53))):
        frames_np = [synth.make_frame(s, 20000) for s in seeds]
        frames = [torch.from_numpy(f).to(dev) for f in frames_np]
        # FIRST HAND (round-4 review item): the forced-active engine against the CPU oracle pipeline itself, not only against
        # the full-map engine
        want, inter = pipeline.run_frames(frames_np, state, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
        res = []
        for act in (True, False):
            eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, B, 20480, dev, active_tiles=act)
            eng.keep_ssfa = True
            eng.set_points(frames)
            eng.tile_cfg.update({"b0.0": 22, "b0.1": 22, "b0.2": 23, "b1.0": 30, "b1.1": 23, "b1.2": 22, "trans_0": 30, "trans_1": 30})
            need = max([int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(2 * B, eng.H, eng.W, 256, sh, 0)) for sh in (0, 1)] +
                       [int(ops.lib.sessd_conv2d_sk_workspace_bytes(B, eng.H, eng.W, 256, 1, 0))])
            eng.sk_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
            if act:
                # (stream-K shape, minimum share) for the Winograd layers, (30, minimum share) = LDS-tiled stream-K kernel,
                # (direct-kernel tile_cfg, 0) for a 1x1 layer / the pair of transposed convs over their lists
                eng.active_cfg = {0: (1, 4), 1: (0, 1), 2: (1, 2), 3: (30, 4), 4: (1, 2), 5: (0, 1), 6: (11, 0), 7: (30, 8), 8: (4, 0)}
                # (the buffer comparison below looks at WHOLE maps: x0 / x1 / tr0 filled everywhere their lists do not compute. The
                # round-5 reach limits for those three -- coarse_fill -- leave the tiles nobody reads alone: checked further down)
                eng.coarse_fill = False
            eng.enqueue()
            out = eng.results()
            x0 = torch.cat([eng.t["x0"].reshape(-1), eng.h["x1"].reshape(-1), eng.t["tr0"].reshape(-1), eng.h["tr1"].reshape(-1),
                            eng.t["mid"].reshape(-1)])
            frac = eng.active_tile_fractions()
            if act:
                assert sorted(eng._active_layers()) == list(range(9))   # every layer really ran over its list
                ssfa = eng.t["out"].cpu()
                assert float((ssfa - inter["ssfa"]).abs().max()) < 5e-4 * max(1.0, float(inter["ssfa"].abs().max()))
                cmp = [_compare_dets(g, w, d) for g, w, d in zip(out, want, inter["debug"])]
                assert all(r["matched"] == r["n"] for r in cmp)
                eng.capture()
                eng.replay()
                out_g = eng.results()
                assert torch.equal(torch.cat([eng.t["x0"].reshape(-1), eng.h["x1"].reshape(-1), eng.t["tr0"].reshape(-1),
                                              eng.h["tr1"].reshape(-1), eng.t["mid"].reshape(-1)]), x0)   # same lists, same shares: same bits
                for a, b in zip(out, out_g):
                    assert np.array_equal(a["box3d_lidar"], b["box3d_lidar"]) and np.array_equal(a["scores"], b["scores"])
                # coarse_fill (round 5): x0 only where b1.0's list can reach, tr0 only inside the pair's listed blocks, x1 not at all
                # (trans_1 walks x1's own list). Poison the three maps: everything downstream must come out the same BITS, and
                # something must have been left alone in each of them
                full = [eng.t["mid"].clone(), eng.h["tr1"].clone(), eng.t["o"].clone(), eng.head.clone()]
                eng.graph = None
                eng.coarse_fill = True
                for m in (eng.t["x0"], eng.h["x1"], eng.t["tr0"]):
                    m.fill_(float("nan"))
                eng.enqueue()
                out_c = eng.results()
                for a_, b_ in zip(full, [eng.t["mid"], eng.h["tr1"], eng.t["o"], eng.head]):
                    assert torch.isfinite(b_).all() and torch.equal(a_, b_)
                for a, b in zip(out, out_c):
                    assert np.array_equal(a["box3d_lidar"], b["box3d_lidar"]) and np.array_equal(a["scores"], b["scores"])
                left = [float(torch.isnan(m).float().mean()) for m in (eng.t["x0"], eng.h["x1"], eng.t["tr0"])]
                print("share of x0 / x1 / tr0 left alone by the reach-limited fill", [round(v, 3) for v in left])
                assert left[0] > 0.15 and left[1] > 0.15 and left[2] > 0.1, left
            res.append((x0, out, frac))
        assert set(res[0][2]) == {"b0.0", "b0.1", "b0.2", "b1.0", "b1.1", "b1.2", "trans_0", "trans_1", "deconv_0+deconv_1"} and res[1][2] == {}
        assert res[0][2]["b1.2"] <= res[0][2]["deconv_0+deconv_1"] < 0.95
        assert 0.05 < res[0][2]["b0.0"] < res[0][2]["b0.1"] < res[0][2]["b0.2"] < 0.6 and res[0][2]["b1.1"] < res[0][2]["b1.2"] < 0.9, res[0][2]
        assert res[0][2]["b1.0"] < res[0][2]["b1.1"] and res[0][2]["trans_0"] == res[0][2]["b0.2"] and res[0][2]["trans_1"] == res[0][2]["b1.2"]
        ref = float(res[1][0].abs().max())
        assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-6 * ref
        for a, b in zip(res[0][1], res[1][1]):
            assert len(a["scores"]) == len(b["scores"])
            assert np.allclose(a["box3d_lidar"], b["box3d_lidar"], rtol=1e-4, atol=1e-4) and np.allclose(a["scores"], b["scores"], rtol=1e-4)
            assert np.array_equal(a["label_preds"], b["label_preds"])


def test_engine_voxelizer_mixed_cap_batch(dev, model):
    """Frames of one batch share the voxelizer workspace and the engine clears it once per batch: a frame that breaks at
    max_voxels must not leak its break index into the next frame (regression: it did)."""
    from oracle import capi
    MV = 4000
    frames = [synth.make_frame(31, 20000), synth.make_frame(32, 20000)[:2500], synth.make_frame(33, 20000)[::7].copy(),
              synth.make_frame(34, 20000)]
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, MV, configs.TEST_CFG, batch_size=4,
                          max_points_per_frame=20480, device=dev, growth=2.0)
    eng.set_points([torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in frames])
    eng.enqueue()
    eng.results()
    prefix = eng.prefix.cpu().numpy()
    hit = []
    for b, f in enumerate(frames):
        v, c, n = capi.points_to_voxel(f, VG["voxel_size"], VG["range"], 5, MV)
        lo, hi = int(prefix[b]), int(prefix[b + 1])
        assert hi - lo == c.shape[0]
        assert np.array_equal(eng.coors[lo:hi, 1:].cpu().numpy(), c)
        assert np.array_equal(eng.nump[lo:hi].cpu().numpy(), n)
        assert np.array_equal(eng.voxels[lo:hi].cpu().numpy(), v)
        hit.append(c.shape[0] == MV)
    assert hit[0] and not hit[1] and not hit[2], hit  # the case the regression needs: capped frame, then uncapped ones


def test_detection_records_follow_the_frames(dev, model):
    """InferenceEngine.attach_records: every enqueue / graph replay appends one fixed-size record per frame on the device
    (what bench.py gathers at the end of the job); unpacked, they equal results() of the same frames."""
    from sessd_hip import dist as sdist
    eng = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 2, 20480, dev)
    rec, cnt = eng.attach_records(8)
    frames = [torch.from_numpy(synth.make_frame(40 + i, 20000 - 500 * i)).to(dev) for i in range(6)]
    want = []
    eng.set_points(frames[0:2]); eng.enqueue(); want += eng.results()
    eng.capture()
    eng.record_cursor.zero_()
    want = []
    for k in range(3):
        eng.set_points(frames[2 * k:2 * k + 2]); eng.replay(); want += eng.results()
    assert int(eng.record_cursor.item()) == 6
    all_rec, all_cnt = sdist.gather_records(rec, cnt, 6)
    got = sdist.unpack_records(all_rec, all_cnt, 6)
    for g, w in zip(got, want):
        assert np.array_equal(g["box3d_lidar"], w["box3d_lidar"]) and np.array_equal(g["scores"], w["scores"])
        assert np.array_equal(g["label_preds"], w["label_preds"])
    assert sum(len(w["scores"]) for w in want) > 20


def test_frames_in_flight_on_cu_sets_equal_the_oracle(dev, model, state):
    """The configuration bench.py times since round 5: four batch-1 engines, two on each half of the chip -- CU-masked streams
    (ops.cu_masked_stream -> hipExtStreamCreateWithCUMask), persistent stream-K launches sized for the half (engine.cu_budget = 128),
    every active-tile layer over its list, captured graphs replayed round-robin with all four in flight. Every frame of every
    engine against the CPU oracle pipeline; replays reproduce the eager bits (fixed launch sizes => fixed summation order)."""
    anchors = pp.create_anchors_3d_range().reshape(-1, 7)
    frames_np = [synth.make_frame(s, 20000) for s in (71, 72, 73, 74, 75, 76, 77, 78)]
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    want, inter = pipeline.run_frames(frames_np, state, VG["range"], VG["voxel_size"], 5, 16000, anchors, None, return_intermediate=True)
    from sessd_hip.runner import engines_on_cu_sets
    engines, streams = engines_on_cu_sets(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, n_engines=4, sets=2, device=dev,
                                          capture=False, max_points_per_frame=20480)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count // 2
    for e in engines:
        assert e.cu_budget == ncu and e._wgs(0) == ncu and e._wgs(1) == 2 * ncu and e._wgs(2) == ncu
        assert sorted(e.active_cfg) == list(range(10)) and all(v[1] == -1 for l, v in e.active_cfg.items() if l in (0, 1, 2, 4, 5))
    eager = []
    for i, f in enumerate(frames[:4]):
        with torch.cuda.stream(streams[i]):
            engines[i].set_points([f])
            engines[i].enqueue()
        streams[i].synchronize()
        eager.append((engines[i].results()[0], engines[i].bev.clone()))
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            e.capture()
    torch.cuda.synchronize()
    got = [None] * 8
    for rnd in range(2):
        for k in range(4):                      # all four in flight before the first is read back
            with torch.cuda.stream(streams[k]):
                engines[k].set_points([frames[rnd * 4 + k]])
                engines[k].replay()
        for k in range(4):
            streams[k].synchronize()
            got[rnd * 4 + k] = engines[k].results()[0]
            if rnd == 0:
                assert torch.equal(engines[k].bev, eager[k][1])
                assert np.array_equal(got[k]["box3d_lidar"], eager[k][0]["box3d_lidar"]) and np.array_equal(got[k]["scores"], eager[k][0]["scores"])
    res = [_compare_dets(g, w, d) for g, w, d in zip(got, want, inter["debug"])]
    assert all(r["matched"] == r["n"] for r in res)
    assert sum(len(g["scores"]) for g in got) > 100
