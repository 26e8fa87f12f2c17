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
import sys
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


def broadcast_state(state, shapes, device, src=0, chunk_elems=256 * 1024 * 1024, matrix_dtype=torch.float32):
    """state: dict name -> tensor on rank `src` (None elsewhere); shapes: the same name -> shape table on every rank.
    Returns name -> tensor ON `device` for every rank (views into a few large flat receive buffers: no host bounce -- the packers of
    HipUNet / HipVAE read them where they landed).  `matrix_dtype=torch.bfloat16` halves the payload of the >= 2-D tensors (Linear /
    conv weights) for the bf16 fast mode: the packers round those to bf16 anyway (RNE, idempotent), so the packed weights are
    bit-identical to rank `src`'s; 1-D tensors (biases, norm scales: kept fp32 by every kernel) always travel in fp32.
    Rank `src` gets the views of its own send buffers, so every rank builds its engine from the same bits."""
    rank = dist.get_rank()
    out = {}
    groups = [([n for n in shapes if len(shapes[n]) >= 2], matrix_dtype), ([n for n in shapes if len(shapes[n]) < 2], torch.float32)]
    for names, dt in groups:
        sizes = [int(math.prod(shapes[n])) for n in names]
        i = 0
        while i < len(names):
            j, tot = i, 0
            while j < len(names) and (tot == 0 or tot + sizes[j] <= chunk_elems):
                tot += sizes[j]
                j += 1
            if rank == src:
                flat = torch.cat([state[n].reshape(-1).to(dt) for n in names[i:j]]).to(device)
            else:
                flat = torch.empty(tot, dtype=dt, device=device)
            dist.broadcast(flat, src=src)
            off = 0
            for n, sz in zip(names[i:j], sizes[i:j]):
                out[n] = flat[off:off + sz].view(shapes[n])
                off += sz
            i = j
    return out


def broadcast_object(obj, src=0):
    """small host objects (config dicts) from rank `src` to everyone"""
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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


def sync_tune_table(src=0, freeze=True):
    """every rank REPLACES its igemm tuning table by rank `src`'s (freefine_amd.ops.tune_table_*) and, with `freeze`, stops
    timing-based tuning on every rank: shapes in the table launch rank `src`'s choice, shapes first seen later take the deterministic
    rule -- identical tile / split-K choices on all ranks either way, hence bit-identical bf16 results across the ranks of a sharded
    run (the tuner picks by timing, which may differ per GPU).  Works on any backend: the table is a small int32 tensor shipped
    with broadcast_object_list.
    The freeze lasts until it is undone: `sync_tune_table.previous` holds the setting it replaced, `restore_tuning()` puts it back, and
    `frozen_tuning()` is the same as a context manager (a later single-GPU phase of the same process can tune again)."""
    from . import ops
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    box = [ops.tune_table_export() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() != src:
        ops.tune_table_clear()
        n = ops.tune_table_import(box[0])
    else:
        n = box[0].shape[0]
    if freeze:
        prev = ops.tune_enable(False)
        if sync_tune_table.previous is None:
            sync_tune_table.previous = prev
        if dist.get_rank() == src:
            print(f"[freefine_amd.dist] igemm tuning table of rank {src} ({n} shapes) installed on {dist.get_world_size()} ranks; timing-based tuning "
                  "frozen (shapes seen later take the deterministic rule) until dist.restore_tuning()", file=sys.stderr, flush=True)
    return n


sync_tune_table.previous = None


def restore_tuning():
    """undo the freeze of sync_tune_table(freeze=True): timing-based tuning back to what it was before"""
    from . import ops
    if sync_tune_table.previous is not None:
        ops.tune_enable(sync_tune_table.previous)
        sync_tune_table.previous = None


class frozen_tuning:
    """with frozen_tuning(src=0): ...  -- the table of rank `src` on every rank and no timing-based tuning inside the block"""

    def __init__(self, src=0):
        self.src = src

    def __enter__(self):
        self.n = sync_tune_table(self.src, freeze=True)
        return self

    def __exit__(self, *exc):
        restore_tuning()
        return False
