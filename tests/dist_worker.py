"""Worker for tests/test_distributed.py: one rank of a gloo world (run as `python tests/dist_worker.py RANK WORLD PORT`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(rank, world_size, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from ctc_attention_mispronunciation_amd import dist as mdist
    assert mdist.world() == (rank, world_size)
    assert mdist.shard_batches(5) == list(range(rank, 5, world_size))
    # posteriors: every rank contributes its own shard, every rank sees all of them in rank order
    lp = torch.full((4, 3, 5), float(rank))
    g = mdist.gather_posteriors(lp)
    assert g.shape == (world_size, 4, 3, 5) and all(float(g[r].mean()) == r for r in range(world_size))
    # decoded ids for 5 batches of 3 utterances; the "decoder" is a deterministic function of the batch content
    batches = [torch.arange(3) + 10 * k for k in range(5)]

    def decode_fn(batch):
        ids = torch.zeros((3, 6), dtype=torch.int32)
        nids = torch.zeros(3, dtype=torch.int32)
        for i, v in enumerate(batch.tolist()):
            n = v % 4 + 1
            ids[i, :n] = v
            nids[i] = n
        return ids, nids
    res = mdist.decode_sharded(batches, decode_fn, pad_len=6)
    flat = [(int(res[k][0][i, 0]), int(res[k][1][i])) for k in range(5) for i in range(3)]
    one = mdist.decode_sharded(batches[:1], decode_fn, pad_len=6)        # fewer batches than ranks: rank 1 has nothing to decode
    assert len(one) == 1 and one[0][0][:, 0].tolist() == [0, 1, 2] and one[0][1].tolist() == [1, 2, 3]
    tot = mdist.sum_counts([rank + 1, 10 * (rank + 1), 0, 0, 0, 0, 0, 7])
    assert tot == [sum(r + 1 for r in range(world_size)), 10 * sum(r + 1 for r in range(world_size)), 0, 0, 0, 0, 0, 7 * world_size]
    # data-parallel gradient averaging of the training step (steps/train_ctc.py): buckets smaller than the model
    from ctc_attention_mispronunciation_amd.steps.train_ctc import allreduce_gradients
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    for k, prm in enumerate(net.parameters()):
        prm.grad = torch.full_like(prm, float((rank + 1) * (k + 1)))
    allreduce_gradients(net, bucket_bytes=64)
    mean_rank = sum(r + 1 for r in range(world_size)) / world_size
    for k, prm in enumerate(net.parameters()):
        assert torch.allclose(prm.grad, torch.full_like(prm, mean_rank * (k + 1))), (k, prm.grad)
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(flat))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
