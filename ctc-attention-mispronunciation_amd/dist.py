"""Utterance-batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The path shards by WHOLE reference batches: with no padding mask anywhere in the reference model
(AA/models/model_ctc.py:186,198,204-205) an utterance's posteriors depend on its batch's padding, so a batch
is the unit that must stay together.  Ranks never exchange anything on the data path; the
only collectives are result gathers: posteriors (what BASELINE.json's north_star names) and/or decoded ids.
"""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_batches(n_batches, rank=None, world_size=None):
    """Indices of the batches this rank decodes: round-robin, so ragged tails spread evenly."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_batches, world_size))


def gather_posteriors(logp, out=None):
    """All-gather equal-shape posteriors [T', B, C] of every rank -> [world, T', B, C] (one collective)."""
    _, w = world()
    if w == 1:
        return logp.unsqueeze(0)
    if out is None:
        out = torch.empty((w,) + tuple(logp.shape), dtype=logp.dtype, device=logp.device)
    dist.all_gather_into_tensor(out.view((-1,) + tuple(logp.shape[1:])), logp.contiguous())   # concatenated along dim 0
    return out


def gather_decoded(ids, nids):
    """All-gather decoded ids [B, T'] int32 and their lengths [B] -> lists of per-rank tensors."""
    _, w = world()
    if w == 1:
        return [ids], [nids]
    all_ids = torch.empty((w,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
    all_n = torch.empty((w,) + tuple(nids.shape), dtype=nids.dtype, device=nids.device)
    dist.all_gather_into_tensor(all_ids.view((-1,) + tuple(ids.shape[1:])), ids.contiguous())
    dist.all_gather_into_tensor(all_n.view(-1), nids.contiguous())
    return list(all_ids.unbind(0)), list(all_n.unbind(0))


def decode_sharded(batches, decode_fn, pad_len):
    """Run `decode_fn(batch) -> (ids [B,pad_len] int32, nids [B] int32)` on this rank's batches and return, on every
    rank, the results of ALL batches in their original order.  Batches must have equal B (pad with empty
    utterances otherwise); a rank with fewer batches contributes zero-length rows for the missing round."""
    rank, w = world()
    mine = shard_batches(len(batches), rank, w)
    rounds = (len(batches) + w - 1) // w
    results = [None] * len(batches)
    proto = None
    for r in range(rounds):
        if r < len(mine):
            ids, nids = decode_fn(batches[mine[r]])
            proto = (ids, nids)
        else:
            if proto is None:
                raise RuntimeError("decode_sharded: a rank without any batch needs equal-shape batches to pad with")
            ids, nids = torch.zeros_like(proto[0]), torch.zeros_like(proto[1])
        assert ids.shape[1] == pad_len
        all_ids, all_n = gather_decoded(ids, nids)
        for src in range(w):
            k = r * w + src
            if k < len(batches):
                results[k] = (all_ids[src], all_n[src])
    return results


def sum_counts(values, device=None):
    """All-reduce (sum) of a rank's evaluation tallies (`steps.test_ctc_nosil.MddCounts.as_list()`, 8 integers): every
    rank gets the totals of the whole job -- the only exchange the batch evaluation needs."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]

