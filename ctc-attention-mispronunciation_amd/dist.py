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


def _all_gather_rows(out, src):
    """out [world * n, ...] <- src [n, ...] of every rank, in rank order.  RCCL ("nccl") gathers device tensors in place
    on the current stream; gloo (CPU tests, and the N>1 rehearsal on a box with fewer GPUs than ranks) has no device
    collectives, so device tensors make the round trip through host memory there."""
    if dist.get_backend() == "gloo" and src.is_cuda:
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, src.contiguous().cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, src.contiguous())


def gather_posteriors(logp, out=None):
    """All-gather equal-shape posteriors [T', B, C] of every rank -> [world, T', B, C] (one collective)."""
    _, w = world()
    if w == 1 and out is None:
        return logp.unsqueeze(0)
    if out is None:
        out = torch.empty((w,) + tuple(logp.shape), dtype=logp.dtype, device=logp.device)
    _all_gather_rows(out.view((-1,) + tuple(logp.shape[1:])), logp)   # concatenated along dim 0
    return out


def gather_decoded(ids, nids, out=None):
    """All-gather decoded ids [B, T'] int32 and their lengths [B] -> lists of per-rank tensors (views of `out` =
    ([world, B, T'], [world, B]) when given)."""
    _, w = world()
    if w == 1 and out is None:
        return [ids], [nids]
    if out is None:
        all_ids = torch.empty((w,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
        all_n = torch.empty((w,) + tuple(nids.shape), dtype=nids.dtype, device=nids.device)
    else:
        all_ids, all_n = out
    _all_gather_rows(all_ids.view((-1,) + tuple(ids.shape[1:])), ids)
    _all_gather_rows(all_n.view(-1), nids)
    return list(all_ids.unbind(0)), list(all_n.unbind(0))


def decode_sharded(batches, decode_fn, pad_len):
    """Run `decode_fn(batch) -> (ids [B,pad_len] int32, nids [B] int32)` on this rank's batches and return, on every
    rank, the results of ALL batches in their original order.  Batches must have equal B (pad with empty
    utterances otherwise); a rank with fewer batches contributes zero-length rows for the missing round."""
    rank, w = world()
    mine = shard_batches(len(batches), rank, w)
    rounds = (len(batches) + w - 1) // w
    results = [None] * len(batches)
    first = decode_fn(batches[mine[0]]) if mine else None
    # Every rank learns B (and the device) before the first gather, so a rank with no batch at all (fewer batches than
    # ranks) can contribute zero rows of the right shape instead of leaving the others waiting in the collective.
    dev = first[0].device if first is not None else (torch.device("cuda", torch.cuda.current_device())
                                                    if w > 1 and dist.get_backend() == "nccl" else torch.device("cpu"))
    nb = torch.tensor([first[0].shape[0] if first is not None else 0], dtype=torch.int64, device=dev)
    if w > 1:
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
    nb = int(nb.item())
    for r in range(rounds):
        if r < len(mine):
            ids, nids = first if r == 0 else decode_fn(batches[mine[r]])
        else:
            ids, nids = torch.zeros((nb, pad_len), dtype=torch.int32, device=dev), torch.zeros((nb,), dtype=torch.int32, device=dev)
        assert ids.shape == (nb, pad_len), "decode_sharded: batches must have equal B and pad_len columns"
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

