"""Multi-GPU support for the GeoBench harness (one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on
ROCm, "gloo" in CPU tests).  The path shards by independent edit cases, exactly like the reference's DistributedSampler
usage (/root/reference/evaluation/FreeFine/freefine_batch_infer_2d.py:141-173, 243-262): no collective inside an edit.

  shard_indices     DistributedSampler(shuffle=False, drop_last=False) semantics: pad by repeating head samples, rank r
                    takes r, r+W, ...
  broadcast_state   rank 0 holds the weights (read from disk once / generated once); every other rank receives them in a few
                    large flat buffers -- large, few messages suit xGMI's point-to-point links (the reference makes each
                    rank read the checkpoint itself, :149)
  gather_results    all_gather_object of the per-rank result dicts (:243), merged by key on every rank
"""
import math

import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    if n == 0:
        return []
    num = math.ceil(n / world)
    total = num * world
    idx = list(range(n))
    pad = total - n
    if pad > 0:
        idx += (idx * math.ceil(pad / n))[:pad]
    return idx[rank:total:world]


def broadcast_state(state, shapes, device, src=0, chunk_elems=256 * 1024 * 1024):
    """state: dict name -> fp32 tensor on rank `src` (None elsewhere); shapes: the same name -> shape table on every rank."""
    names = list(shapes)
    sizes = [int(torch.tensor(shapes[n]).prod()) for n in names]
    out = {}
    rank = dist.get_rank()
    i = 0
    while i < len(names):
        j, tot = i, 0
        while j < len(names) and (tot == 0 or tot + sizes[j] <= chunk_elems):
            tot += sizes[j]
            j += 1
        if rank == src:
            flat = torch.cat([state[n].reshape(-1).float() for n in names[i:j]]).to(device)
        else:
            flat = torch.empty(tot, dtype=torch.float32, device=device)
        dist.broadcast(flat, src=src)
        off = 0
        flat_cpu = flat.cpu()
        for n, s in zip(names[i:j], sizes[i:j]):
            out[n] = flat_cpu[off:off + s].reshape(shapes[n]).clone()
            off += s
        i = j
    return out


def gather_results(local_results):
    """list[dict] per rank -> merged list (duplicates from sampler padding removed by `key`)."""
    world = dist.get_world_size()
    gathered = [None] * world
    dist.all_gather_object(gathered, local_results)
    merged, seen = [], set()
    for part in gathered:
        for item in part:
            k = item.get("key") if isinstance(item, dict) else None
            if k is not None:
                if k in seen:
                    continue
                seen.add(k)
            merged.append(item)
    return merged


def sync_tune_table(src=0):
    """every rank adopts rank `src`'s igemm tuning table (freefine_amd.ops.tune_table_*): identical tile / split-K choices on all
    ranks, hence bit-identical bf16 results across the ranks of a sharded run (the tuner picks by timing, which may differ per GPU).
    Works on any backend: the table is a small int32 tensor shipped with broadcast_object_list."""
    from . import ops
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    box = [ops.tune_table_export() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return ops.tune_table_import(box[0]) if dist.get_rank() != src else box[0].shape[0]
