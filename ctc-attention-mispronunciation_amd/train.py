"""Training-side plumbing over the C ABI (include/mdd_hip.h, "SURVEY 8(f) #3"): the pieces `run_epoch` of the reference uses
(AA/steps/train_ctc.py:28-105) -- train-mode ``model(inputs, trans)``, ``nn.CTCLoss(reduction='sum')``, ``loss.backward()``,
``optimizer.step()`` -- as torch.autograd Functions and an optimizer whose arithmetic runs in libmdd_hip.so.

* ``TrainHandle``   one mdd_train_ws: tensor table, forward / backward on raw device pointers.
* ``model_forward_train(model, x, x1, masks=None)``   what ``CTC_Model.forward`` calls in train mode; differentiable w.r.t. every
  parameter (autograd Function; the model's BatchNorm running statistics are updated in place, as nn.BatchNorm does).
* ``CTCLoss``       drop-in for ``nn.CTCLoss`` (same constructor / call signature, train_ctc.py:72,186) over mdd_ctc_loss.
* ``Adam``          ``torch.optim.Adam``-compatible optimizer (lr, betas, eps, weight_decay, state_dict) over mdd_adam_step.
PyTorch supplies device memory, streams and the autograd tape; no torch operator computes anything on this path.
"""
import ctypes as C

import torch

from . import _lib


class TrainHandle(object):
    def __init__(self, cfg, device_index):
        _lib.require_gpu()
        self.handle = C.c_void_p()
        self.device_index = device_index
        _lib.check(_lib.lib().mdd_train_create(C.byref(cfg), device_index, C.byref(self.handle)))
        L = _lib.lib()
        n = L.mdd_train_num_tensors(self.handle)
        self.keys, self.numel, self.is_buffer = [], [], []
        buf = C.create_string_buffer(128)
        for i in range(n):
            ne, ib = C.c_int64(0), C.c_int32(0)
            _lib.check(L.mdd_train_tensor_info(self.handle, i, buf, 128, C.byref(ne), C.byref(ib)))
            self.keys.append(buf.value.decode()); self.numel.append(ne.value); self.is_buffer.append(bool(ib.value))
        self.n_masks = L.mdd_train_num_masks(self.handle)

    def mask_shapes(self, geom_channels, hidden, B, T, W1, W2):
        return [(B, geom_channels, T, W1), (B, geom_channels, T // 2, W2)] + [(T // 2, B, 2 * hidden)] * (self.n_masks - 2)

    @staticmethod
    def _ptr_array(tensors):
        arr = (C.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr() if t is not None else None
        return arr

    def forward(self, tensors, x, x1, masks, seed, p_drop):
        B, T, _ = x.shape
        out = torch.empty((T // 2, B, self._num_class(tensors)), dtype=torch.float32, device=x.device)
        marr = None
        if masks is not None:
            assert len(masks) == self.n_masks
            marr = self._ptr_array(masks)
        _lib.check(_lib.lib().mdd_train_forward(self.handle, self._ptr_array(tensors), C.c_void_p(x.data_ptr()), B, T, C.c_void_p(x1.data_ptr()),
                                                x1.shape[1], marr, C.c_uint64(seed), C.c_float(p_drop), C.c_void_p(out.data_ptr()),
                                                _lib.current_stream_ptr()))
        return out

    def _num_class(self, tensors):
        return tensors[self.keys.index("fc.1.weight")].shape[0]

    def backward(self, tensors, dlogp, grads):
        _lib.check(_lib.lib().mdd_train_backward(self.handle, self._ptr_array(tensors), C.c_void_p(dlogp.data_ptr()), self._ptr_array(grads),
                                                 _lib.current_stream_ptr()))

    def sync(self):
        if _lib.lib().mdd_train_sync(self.handle, _lib.current_stream_ptr()) != 0:
            raise IndexError(_lib.lib().mdd_last_error().decode())

    def close(self):
        if self.handle:
            _lib.lib().mdd_train_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _TrainForward(torch.autograd.Function):
    """logp = CTC_Model.forward(x, x1) in train mode; backward hands every parameter its gradient."""

    @staticmethod
    def forward(ctx, handle, x, x1, masks, seed, p_drop, all_tensors, *params):
        # all_tensors: the state tensors in the handle's order (parameters AND running-statistics buffers); params are the
        # differentiable ones among them (passed separately so that autograd tracks them)
        ctx.handle, ctx.all_tensors, ctx.keep = handle, all_tensors, (x, x1, masks)
        with torch.cuda.device(x.device):
            return handle.forward(all_tensors, x, x1, masks, seed, p_drop)

    @staticmethod
    def backward(ctx, dlogp):
        h = ctx.handle
        dev = dlogp.device
        grads = [None if h.is_buffer[i] else torch.empty(h.numel[i], dtype=torch.float32, device=dev) for i in range(len(h.keys))]
        with torch.cuda.device(dev):
            h.backward(ctx.all_tensors, dlogp.contiguous(), grads)
        out = [g.view_as(t) for g, t, b in zip(grads, ctx.all_tensors, h.is_buffer) if not b]
        return (None, None, None, None, None, None, None) + tuple(out)


def model_forward_train(model, x, x1, masks=None, seed=None):
    """Train-mode forward of the drop-in CTC_Model through libmdd_hip (called by CTC_Model.forward when model.training)."""
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    xd = x.to(dev, torch.float32).contiguous()
    idd = x1.to(dev, torch.int64).contiguous()
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    h = getattr(model, "_train_handle", None)
    if h is None or h.device_index != index:
        h = TrainHandle(model._config(), index)
        model._train_handle = h
    want = getattr(model, "train_precision", None)          # None: the library default (exact fp32, or MDD_TRAIN_PRECISION)
    if want is not None and want != getattr(h, "precision", None):
        _lib.check(_lib.lib().mdd_train_set_precision(h.handle, {"f32": 0, "bf16x3": 1}[want]))
        h.precision = want
    sd = dict(model.named_parameters())
    sd.update(dict(model.named_buffers()))
    tensors = []
    for key, ne in zip(h.keys, h.numel):
        t = sd[key]
        if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("train-mode forward needs the model on %s in float32 (model.to(%r)); %s is on %s" % (dev, str(dev), key, t.device))
        assert t.numel() == ne, (key, t.numel(), ne)
        tensors.append(t)
    params = [t for t, b in zip(tensors, h.is_buffer) if not b]
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # consumes torch's CPU generator: reproducible under torch.manual_seed
    if masks is not None:
        masks = [m.to(dev, torch.uint8).contiguous() for m in masks]
    out = _TrainForward.apply(h, xd, idd, masks, seed, float(model.drop_out), tensors, *params)
    for name, buf in model.named_buffers():                        # nn.BatchNorm counts its batches
        if name.endswith("num_batches_tracked"):
            buf += 1
    if getattr(model, "strict_errors", True):
        h.sync()
    model._dirty = True                                            # the eval path's packed weight copies are stale now
    return out if x.device == dev else out.to(x.device)


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank):
        from .hip_model import ctc_loss
        nll, grad = ctc_loss(log_probs, targets, input_lengths, target_lengths, blank=blank, want_grad=log_probs.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0))
        return nll

    @staticmethod
    def backward(ctx, gnll):
        (grad,) = ctx.saved_tensors
        return grad * gnll.view(1, -1, 1), None, None, None, None


class CTCLoss(torch.nn.Module):
    """``nn.CTCLoss`` over the gfx950 lattice kernel (mdd_ctc_loss): log_probs [T,B,C] (cuda, float32), padded 2-D or
    concatenated 1-D targets, input / target lengths -- same call and reductions as torch (train_ctc.py:72,186 uses
    reduction='sum', blank 0)."""

    def __init__(self, blank=0, reduction="mean", zero_infinity=False):
        super(CTCLoss, self).__init__()
        if reduction not in ("none", "mean", "sum"):
            raise ValueError("%s is not a valid value for reduction" % reduction)
        self.blank, self.reduction, self.zero_infinity = blank, reduction, zero_infinity

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        dev = log_probs.device
        il = torch.as_tensor(input_lengths, dtype=torch.int64)
        tl = torch.as_tensor(target_lengths, dtype=torch.int64)
        tg = torch.as_tensor(targets)
        if tg.dim() == 1:                                           # concatenated targets -> padded [B, Lmax]
            B, Lm = tl.numel(), int(tl.max()) if tl.numel() else 0
            pad = torch.zeros((B, max(Lm, 1)), dtype=torch.int64)
            off = 0
            tgc = tg.cpu()
            for b in range(B):
                n = int(tl[b])
                pad[b, :n] = tgc[off:off + n]
                off += n
            tg = pad
        nll = _CTCLossFn.apply(log_probs, tg.to(dev), il.to(dev), tl.to(dev), self.blank)
        if self.zero_infinity:
            nll = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
        if self.reduction == "none":
            return nll
        if self.reduction == "sum":
            return nll.sum()
        return (nll / tl.to(dev).clamp(min=1).to(nll.dtype)).mean()


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's update (no amsgrad) with the arithmetic in mdd_adam_step: g += weight_decay * p; biased first and
    second moments; bias-corrected step.  State keys ('step', 'exp_avg', 'exp_avg_sq') follow torch's, so a state_dict saved
    by either loads into the other (save_package keeps 'optim_dict', model_ctc.py:262)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
            steps = {int(self.state[p]["step"]) for p in ps}
            for step in steps:                                       # parameters added later may be at a different step
                grp = [p for p in ps if int(self.state[p]["step"]) == step]
                for p in grp:
                    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                        raise RuntimeError("mdd Adam: parameters and gradients must be contiguous float32 CUDA tensors")
                n = len(grp)
                arr = lambda ts: TrainHandle._ptr_array(ts)           # noqa: E731
                numel = (C.c_int64 * n)(*[p.numel() for p in grp])
                with torch.cuda.device(grp[0].device):
                    _lib.check(L.mdd_adam_step(arr(grp), arr([p.grad for p in grp]), arr([self.state[p]["exp_avg"] for p in grp]),
                                               arr([self.state[p]["exp_avg_sq"] for p in grp]), numel, n, step, C.c_float(group["lr"]),
                                               C.c_float(group["betas"][0]), C.c_float(group["betas"][1]), C.c_float(group["eps"]),
                                               C.c_float(group["weight_decay"]), _lib.current_stream_ptr()))
        return loss
