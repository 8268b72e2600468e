"""The training loop around the hot path -- mirror of the reference's ``steps/train_ctc.py`` (AA/steps/train_ctc.py):

* ``run_epoch`` (:28-105)  one pass over a data loader: train-mode forward, ``loss_fn(out, targets, (frac * T').long(), target_sizes)
  / batch_size``, greedy frame error count through ``model.compute_wer``, ``zero_grad / backward / step``;  same arguments and
  ``(1 - error rate, mean loss)`` return.  With ``torch.distributed`` initialised, gradients are averaged over the ranks before the
  optimizer step (data-parallel config 5 of BASELINE.json: every rank steps its own 32-utterance shard; BatchNorm statistics stay
  per rank, the reference has no SyncBN).
* ``LrSchedule`` (:207-268)  the dev-loss driven halving the reference writes inline in ``main()``: a new best (by more than
  ``end_adjust_acc``) resets the patience; ten epochs inside the band, or one outside it, restore the best state, multiply the rate
  by ``decay`` and count an adjustment; eight adjustments stop training.
* ``build_training(...)``  model + ``CTCLoss(reduction='sum')`` + ``Adam(lr, weight_decay)`` as :184-187 wires them.
The arithmetic of every step runs in libmdd_hip (train.py); this file is host control flow only.
"""
import copy

import torch

from ..train import Adam, CTCLoss


def allreduce_gradients(model, bucket_bytes=64 << 20):
    """Average the gradients over the ranks (RCCL all-reduce over xGMI on ROCm; no-op without an initialised process group).
    Gradients are packed into buckets of ~64 MB so the ring moves few large messages (21 M parameters = 85 MB: two buckets)."""
    import torch.distributed as dist
    import os
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not os.environ.get("MDD_FORCE_DIST")):
        return                                       # (MDD_FORCE_DIST: rehearse the collective path on a one-rank group)
    world = dist.get_world_size()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        if dist.get_backend() == "gloo" and flat.is_cuda:
            h = flat.cpu()
            dist.all_reduce(h)
            flat.copy_(h)
        else:
            dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for g in grads:
        bucket.append(g)
        size += g.numel() * 4
        if size >= bucket_bytes:
            flush()
            bucket, size = [], 0
    flush()


def run_epoch(epoch_id, model, data_iter, loss_fn, device, optimizer=None, print_every=20, is_training=True):
    """AA/steps/train_ctc.py:28-105, same arguments, prints and return value.  The reference pulls the loss (``loss.item()``) and the
    frame argmax (``torch.max`` + ``compute_wer`` on numpy copies) to the host on EVERY step; at 14-33 ms per HIP step those two
    synchronisations are no longer free, so the bookkeeping stays on the device: the loss is accumulated in a float64 device
    scalar (the same sum, in the same order, as the reference's Python float), the greedy prediction of ``compute_wer``
    (argmax, drop repeats, drop blanks: model_ctc.py:227-244) is mdd_greedy's output, and both come to the host once per print
    interval, where the edit distances of all pending utterances are one native call (mdd_align_batch)."""
    import numpy as np
    from ..utils.ctcDecoder import GreedyDecoder, align_ids_batch
    model.train() if is_training else model.eval()
    total_loss, total_tokens, total_errs, i = 0, 0, 0, -1
    greedy = GreedyDecoder({}, space_idx=-1, blank_index=0)
    loss_acc, pending = None, []

    def flush():
        nonlocal total_loss, total_tokens, total_errs, loss_acc, pending
        if loss_acc is not None:
            total_loss = float(loss_acc.item())               # the one host synchronisation of the interval
        for ids, n, tg, tl in pending:
            ids, n, tg, tl = ids.cpu().numpy(), n.cpu().numpy(), tg.cpu().numpy().astype(np.int32), tl.cpu().numpy().astype(np.int32)
            dist = align_ids_batch(tg, tl, ids, n)[0]
            empty = dist < 0                                  # ed.eval with an empty side = the length of the other one
            total_errs += int(dist[~empty].sum()) + int(np.maximum(tl, n)[empty].sum())
            total_tokens += int(tl.sum())
        pending = []
    for i, data in enumerate(data_iter):
        inputs, input_sizes, targets, target_sizes, trans, trans_sizes, utt_list = data
        inputs, input_sizes = inputs.to(device), input_sizes.to(device)
        targets, target_sizes, trans = targets.to(device), target_sizes.to(device), trans.to(device)
        with torch.set_grad_enabled(is_training):
            out = model(inputs, trans)
            out_len, batch_size, _ = out.size()
            input_sizes = (input_sizes * out_len).long()
            loss = loss_fn(out, targets, input_sizes, target_sizes)
            loss = loss / batch_size
        if out.is_cuda:
            loss_acc = loss.detach().double() if loss_acc is None else loss_acc + loss.detach().double()
            ids, n = greedy.decode_ids(out.detach(), input_sizes)
            pending.append((ids, n, targets, target_sizes))
        else:                                                 # (CPU inputs: the forward has copied the posteriors back already)
            total_loss += loss.item()
            _, index = torch.max(out.detach(), dim=-1)
            batch_errs, batch_tokens = model.compute_wer(index.transpose(0, 1).cpu().numpy(), input_sizes.cpu().numpy(), targets.cpu().numpy(),
                                                         target_sizes.cpu().numpy())
            total_errs += batch_errs
            total_tokens += batch_tokens
        if (i + 1) % print_every == 0 and is_training:
            flush()
            print('Epoch = %d, step = %d, total_loss = %.4f, total_wer = %.4f' % (epoch_id, i + 1, total_loss / (i + 1), total_errs / total_tokens))
        if is_training:
            optimizer.zero_grad()
            loss.backward()
            allreduce_gradients(model)
            optimizer.step()
    flush()
    average_loss = total_loss / (i + 1)
    print("Epoch %d %s done, total_loss: %.4f, total_wer: %.4f" % (epoch_id, "Train" if is_training else "Valid", average_loss, total_errs / total_tokens))
    return 1 - total_errs / total_tokens, average_loss


class LrSchedule(object):
    """State machine of train_ctc.py:207-268.  Call ``begin_epoch()`` before an epoch (applies a pending decay to the optimizer,
    returns False when training must stop) and ``end_epoch(acc, dev_loss)`` after its validation pass."""

    def __init__(self, model, optimizer, init_lr, decay, end_adjust_acc, num_epoches):
        self.model, self.optimizer = model, optimizer
        self.learning_rate, self.decay, self.band, self.num_epoches = init_lr, decay, end_adjust_acc, num_epoches
        self.count = 0
        self.loss_best = self.loss_best_true = 1000
        self.adjust_rate_flag = self.stop_train = False
        self.adjust_time = 0
        self.adjust_rate_count = 0
        self.acc_best = 0
        self.model_state = self.op_state = self.best_model_state = self.best_op_state = None

    def begin_epoch(self):
        if self.stop_train or self.count >= self.num_epoches:
            return False
        self.count += 1
        if self.adjust_rate_flag:
            self.learning_rate *= self.decay
            self.adjust_rate_flag = False
            for group in self.optimizer.param_groups:
                group['lr'] *= self.decay
        return True

    def _snapshot(self):
        return copy.deepcopy(self.model.state_dict()), copy.deepcopy(self.optimizer.state_dict())

    def end_epoch(self, acc, dev_loss):
        if dev_loss < (self.loss_best - self.band):
            self.loss_best = self.loss_best_true = dev_loss
            self.adjust_rate_count = 0
            self.model_state, self.op_state = self._snapshot()
        elif dev_loss < self.loss_best + self.band:
            self.adjust_rate_count += 1
            if dev_loss < self.loss_best and dev_loss < self.loss_best_true:
                self.loss_best_true = dev_loss
                self.model_state, self.op_state = self._snapshot()
        else:
            self.adjust_rate_count = 10
        if acc > self.acc_best:
            self.acc_best = acc
            self.best_model_state, self.best_op_state = self._snapshot()
        if self.adjust_rate_count == 10:
            self.adjust_rate_flag = True
            self.adjust_time += 1
            self.adjust_rate_count = 0
            if self.loss_best > self.loss_best_true:
                self.loss_best = self.loss_best_true
            self.model.load_state_dict(self.model_state)
            self.optimizer.load_state_dict(self.op_state)
        if self.adjust_time == 8:
            self.stop_train = True

    def finish(self):
        """Load the best-accuracy state (train_ctc.py:283-285) before the checkpoint is written."""
        if self.best_model_state is not None:
            self.model.load_state_dict(self.best_model_state)
            self.optimizer.load_state_dict(self.best_op_state)


def build_training(model, init_lr=1e-3, weight_decay=5e-4):
    """(loss_fn, optimizer) as train_ctc.py:186-187 creates them."""
    return CTCLoss(reduction='sum'), Adam(model.parameters(), lr=init_lr, weight_decay=weight_decay)
