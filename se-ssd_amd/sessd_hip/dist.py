"""Multi-GPU layer of the inference path: frames shard across ranks, nothing else is exchanged.

Mirrors the reference's data-parallel evaluation (tools/dist_test.py:60-186: DistributedSampler(shuffle=False) +
all_gather of pickled per-rank detection dicts, det3d/torchie/trainer/utils.py:115-155). One process per GPU over
torch.distributed (backend "nccl" == RCCL over xGMI on MI355X nodes, "gloo" in the CPU tests). There is no
data-path collective: the only communication is ONE fixed-size all_gather of (<=100 x 9 float) records per frame
at the end (instead of padded pickle byte tensors), or none at all when every rank writes its own results."""
import math

import torch
import torch.distributed as dist


def collectives_enabled(group=None):
    """True when a collective must really be issued: a process group exists and has more than one rank -- or has ONE rank and
    SESSD_FORCE_COLLECTIVES=1 is set, which turns every world-1 short-circuit of this package off (gather_records, allreduce_flat,
    the SyncBN statistics all-reduce, bench.py's barriers): the way the RCCL path is executed on a box with a single GPU
    (tests/test_rccl_gpu.py; review item: no RCCL call of this project had ever run before an 8-GPU node)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("SESSD_FORCE_COLLECTIVES") == "1"


def shard_indices(num_frames, rank, world_size):
    """DistributedSampler(shuffle=False) semantics (det3d/datasets/loader/sampler.py:74-96): the index list is padded
    by wrapping to a multiple of world_size, rank r takes indices r, r+world, ... Returns (indices, num_padded)."""
    per = int(math.ceil(num_frames / float(world_size)))
    total = per * world_size
    idx = list(range(num_frames))
    idx += idx[: total - num_frames]
    return idx[rank:total:world_size], total - num_frames


def pack_detections(dets, post_max=100):
    """list of per-frame dicts(box3d_lidar (n,7), scores (n,), label_preds (n,)) -> (F, post_max, 9) float32 + counts (F,)."""
    F = len(dets)
    rec = torch.zeros((F, post_max, 9), dtype=torch.float32)
    cnt = torch.zeros((F,), dtype=torch.int32)
    for f, d in enumerate(dets):
        n = int(len(d["scores"]))
        cnt[f] = n
        if n:
            rec[f, :n, :7] = torch.as_tensor(d["box3d_lidar"], dtype=torch.float32)
            rec[f, :n, 7] = torch.as_tensor(d["scores"], dtype=torch.float32)
            rec[f, :n, 8] = torch.as_tensor(d["label_preds"]).to(torch.float32)
    return rec, cnt


def gather_detections(local_dets, num_frames, post_max=100, device=None):
    """All ranks call this with the detections of THEIR shard (in shard order). Returns, on every rank, the list of
    `num_frames` per-frame dicts in dataset order (padding duplicates dropped), via one all_gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per = int(math.ceil(num_frames / float(world)))
    assert len(local_dets) == per, "every rank must process ceil(num_frames/world) frames (padded shard)"
    rec, cnt = pack_detections(local_dets, post_max)
    if device is not None:
        rec, cnt = rec.to(device), cnt.to(device)
    if world > 1:
        recs = [torch.empty_like(rec) for _ in range(world)]
        cnts = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(recs, rec)
        dist.all_gather(cnts, cnt)
    else:
        recs, cnts = [rec], [cnt]
    out = [None] * num_frames
    for r in range(world):
        idx, _ = shard_indices(num_frames, r, world)
        for j, f in enumerate(idx):
            if out[f] is None:
                n = int(cnts[r][j])
                a = recs[r][j, :n].cpu()
                out[f] = dict(box3d_lidar=a[:, :7].numpy(), scores=a[:, 7].numpy(), label_preds=a[:, 8].long().numpy())
    return out


def gather_records(records, counts, num_frames):
    """Device-side twin of gather_detections for the engine's records (InferenceEngine.attach_records): every rank passes its
    (per, post_max, 9) float32 records + (per,) int32 counts in shard order (per = ceil(num_frames / world)); ONE all_gather each
    (RCCL over xGMI: 3.6 KB per frame). Returns (world, per, post_max, 9) and (world, per) tensors on the calling device; nothing
    touches the host."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    per = int(math.ceil(num_frames / float(world)))
    assert records.shape[0] >= per and counts.shape[0] >= per
    rec, cnt = records[:per].contiguous(), counts[:per].contiguous()
    if not collectives_enabled():
        return rec.unsqueeze(0), cnt.unsqueeze(0)
    # concatenated along dim 0 (the layout both the RCCL and the gloo backend accept), viewed as (world, per, ...)
    all_rec = torch.empty((world * per,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    all_cnt = torch.empty((world * per,), dtype=cnt.dtype, device=cnt.device)
    dist.all_gather_into_tensor(all_rec, rec)
    dist.all_gather_into_tensor(all_cnt, cnt)
    return all_rec.view((world, per) + tuple(rec.shape[1:])), all_cnt.view(world, per)


def unpack_records(all_rec, all_cnt, num_frames):
    """(world, per, post_max, 9) + (world, per) -> list of num_frames per-frame dicts in dataset order (padding duplicates dropped)."""
    world = all_rec.shape[0]
    rec, cnt = all_rec.cpu(), all_cnt.cpu()
    out = [None] * num_frames
    for r in range(world):
        idx, _ = shard_indices(num_frames, r, world)
        for j, f in enumerate(idx):
            if out[f] is None:
                n = int(cnt[r, j])
                a = rec[r, j, :n]
                out[f] = dict(box3d_lidar=a[:, :7].numpy(), scores=a[:, 7].numpy(), label_preds=a[:, 8].long().numpy())
    return out
