"""The CU-masked streams behind bench.py's headline (round 5: four frames in flight on two CU sets; ops.cu_masked_stream over
hipExtStreamCreateWithCUMask) really CONFINE kernels -- eagerly and when a graph that was captured on torch's own capture stream is
replayed on the masked stream, which is how every engine runs (round-5 advisor finding: results were tested, confinement was not).
sessd_debug_cu_probe records the physical compute unit (XCC, shader engine, shader array, CU) of each of a few thousand one-wave
workgroups that spin long enough to be spread over every CU their queue may use."""
import numpy as np
import pytest
import torch

from sessd_hip import ops
from sessd_hip._lib import check, lib

pytestmark = pytest.mark.gpu


def _probe(stream, n=4096, spin=40000):
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        check(lib.sessd_debug_cu_probe(ids.data_ptr(), n, spin, torch.cuda.current_stream().cuda_stream), "debug_cu_probe")
    stream.synchronize()
    return _cus(ids)


def _cus(ids):
    v = ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    return set((v & ~np.int64(0xFF)).tolist())    # drop wave / SIMD / pipe: (xcc << 16) | se | sh | cu


def test_cu_sets_are_disjoint_and_cover_the_chip(dev):
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    plain = _probe(torch.cuda.Stream(device=dev))
    assert len(plain) == total, (len(plain), total)               # the probe sees every CU of an unmasked stream
    for parts in (2, 4):
        sets = [_probe(ops.cu_masked_stream(k, parts, dev)[0]) for k in range(parts)]
        for k, s in enumerate(sets):
            assert len(s) == total // parts, (parts, k, len(s))   # exactly its share of the CUs ...
            assert s <= plain
            for q in range(k):
                assert not (s & sets[q]), (parts, k, q)           # ... and none of another set's
        assert set().union(*sets) == plain
    # how the contiguous halves lie on the eight accelerator dies (recorded, not asserted beyond the totals)
    half = _probe(ops.cu_masked_stream(0, 2, dev)[0])
    per_xcc = np.bincount([k >> 16 for k in half], minlength=8)
    print("CUs of set 0 per XCC:", per_xcc.tolist())
    assert per_xcc.sum() == total // 2


def test_graph_replay_on_a_masked_stream_stays_on_its_cu_set(dev):
    """what InferenceEngine.capture() / replay() do: capture under torch.cuda.graph (torch's capture stream, unmasked), replay with the
    masked stream current"""
    sets, graphs, outs = [], [], []
    for k in range(2):
        st, ncu = ops.cu_masked_stream(k, 2, dev)
        sets.append(_probe(st))
        ids = torch.zeros(4096, dtype=torch.int32, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g):
                check(lib.sessd_debug_cu_probe(ids.data_ptr(), 4096, 40000, torch.cuda.current_stream().cuda_stream), "debug_cu_probe")
            ids.zero_()
            g.replay()
        st.synchronize()
        graphs.append((g, ids, st))
        outs.append(_cus(ids))
        assert len(sets[k]) == ncu
    for k in range(2):
        assert outs[k] == sets[k], (k, len(outs[k]), len(outs[k] - sets[k]))   # the replayed kernel ran on the stream's CUs, all of them
    assert not (outs[0] & outs[1])
    # both graphs replayed at the same time on their own streams: still their own sets
    for g, ids, st in graphs:
        ids.zero_()
    torch.cuda.synchronize()
    for g, ids, st in graphs:
        with torch.cuda.stream(st):
            g.replay()
    torch.cuda.synchronize()
    for k, (g, ids, st) in enumerate(graphs):
        assert _cus(ids) <= sets[k], k


def test_engine_refuses_unmasked_side_branches_on_a_cu_set(dev):
    """fork_front / fork_active run part of a frame on a side stream of the engine's own (plain, whole chip): with a CU budget that
    would leave the set -- the engine refuses instead (round-5 advisor finding)"""
    from sessd_hip import configs
    from sessd_hip.engine import InferenceEngine
    VG = configs.VOXEL_GENERATOR
    model = configs.build_synthetic_detector(dev, seed=0)
    e = InferenceEngine(model, VG["range"], VG["voxel_size"], 5, 16000, configs.TEST_CFG, 1, 20480, dev)
    e.cu_budget = 128
    e.fork_front = True
    with pytest.raises(RuntimeError, match="CU set"):
        e.enqueue()
    e.fork_front, e.fork_active = False, True
    with pytest.raises(RuntimeError, match="CU set"):
        e.enqueue()
