"""Thin object over the C ABI for callers that hold a state_dict of numpy arrays / tensors and want
``forward`` without constructing torch modules (bench.py, tests, the sharded driver).  The
reference-shaped drop-in class is ``models.model_ctc.CTC_Model``; both call the same entry points."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class HipModel(object):
    def __init__(self, geom, state_dict, device=0, taps=False, precision=None):
        _lib.require_gpu()
        self.geom = geom
        self.device = torch.device("cuda", device)
        self.handle = C.c_void_p()
        cfg = _lib.MddConfig(feat=geom.feat, hidden=geom.hidden, layers=geom.layers, num_class=geom.num_class,
                             channels=geom.channels, emb_rows=geom.emb_rows, emb_dim=geom.emb_dim, bn_eps=1e-5)
        _lib.check(_lib.lib().mdd_create(C.byref(cfg), device, C.byref(self.handle)))
        if precision is not None:      # 'f32x6' (fp32-grade on the bf16 matrix cores: the default), 'f32' (exact fp32 MFMA) or 'bf16x3' (flagged variant)
            _lib.check(_lib.lib().mdd_set_precision(self.handle, {'f32': 0, 'bf16x3': 1, 'f32x6': 2}[precision]))
        self.load_state_dict(state_dict)
        if taps:
            _lib.check(_lib.lib().mdd_enable_taps(self.handle, 1))

    @property
    def precision(self):
        return {0: 'f32', 1: 'bf16x3', 2: 'f32x6'}[_lib.lib().mdd_get_precision(self.handle)]

    def load_state_dict(self, state_dict):
        L = _lib.lib()
        for key, val in state_dict.items():
            a = val.detach().cpu().numpy() if torch.is_tensor(val) else np.asarray(val)
            if a.dtype.kind != "f":
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(L.mdd_load_weight(self.handle, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        _lib.check(L.mdd_finalize_weights(self.handle))

    def forward(self, x, x1, out=None, sync_errors=False):
        """x [B,T,F] f32 cuda, x1 [B,L] i64 cuda -> logp [T/2,B,C] (enqueued on the current stream)."""
        assert x.is_cuda and x1.is_cuda and x.dtype == torch.float32 and x1.dtype == torch.int64
        x, x1 = x.contiguous(), x1.contiguous()
        B, T, _ = x.shape
        if out is None:
            out = torch.empty((T // 2, B, self.geom.num_class), dtype=torch.float32, device=x.device)
        st = _lib.current_stream_ptr()
        _lib.check(_lib.lib().mdd_forward(self.handle, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(x1.data_ptr()),
                                          x1.shape[1], C.c_void_p(out.data_ptr()), st))
        if sync_errors:
            if _lib.lib().mdd_sync(self.handle, st) != 0:
                raise IndexError(_lib.lib().mdd_last_error().decode())
        return out

    def forward_fused(self, x, x1, frames, canon, out=None, sync_errors=False):
        """Several reference batches of different padded lengths in one launch sequence (mdd_forward_fused): x [B,T,F] with every
        batch zero-padded to the common T, x1 [B,L]; frames [B] int32 = T_g/2 and canon [B] int32 = L_g of each row's own batch.
        Rows t < frames[b] of the result equal forward() on that batch alone, bit for bit."""
        assert x.is_cuda and x1.is_cuda and x.dtype == torch.float32 and x1.dtype == torch.int64
        assert frames.is_cuda and canon.is_cuda and frames.dtype == torch.int32 and canon.dtype == torch.int32
        x, x1, frames, canon = x.contiguous(), x1.contiguous(), frames.contiguous(), canon.contiguous()
        B, T, _ = x.shape
        if out is None:
            out = torch.empty((T // 2, B, self.geom.num_class), dtype=torch.float32, device=x.device)
        st = _lib.current_stream_ptr()
        _lib.check(_lib.lib().mdd_forward_fused(self.handle, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(x1.data_ptr()), x1.shape[1],
                                                C.c_void_p(frames.data_ptr()), C.c_void_p(canon.data_ptr()), C.c_void_p(out.data_ptr()), st))
        if sync_errors:
            if _lib.lib().mdd_sync(self.handle, st) != 0:
                raise IndexError(_lib.lib().mdd_last_error().decode())
        return out

    def forward_raw(self, raw, x1, out=None, sync_errors=False):
        """raw [B,T_raw,F/3] f32 cuda (unstacked frames), x1 [B,L] i64 cuda -> logp, exactly as
        forward(stack_features(raw), x1): the stack/skip of data_loader.py:138-142 is applied on the fly."""
        assert raw.is_cuda and x1.is_cuda and raw.dtype == torch.float32 and x1.dtype == torch.int64
        raw, x1 = raw.contiguous(), x1.contiguous()
        B, T_raw, D = raw.shape
        assert 3 * D == self.geom.feat
        T = _lib.lib().mdd_stack_len(T_raw, 2, 2)
        if out is None:
            out = torch.empty((T // 2, B, self.geom.num_class), dtype=torch.float32, device=raw.device)
        st = _lib.current_stream_ptr()
        _lib.check(_lib.lib().mdd_forward_raw(self.handle, C.c_void_p(raw.data_ptr()), B, T_raw, C.c_void_p(x1.data_ptr()),
                                              x1.shape[1], C.c_void_p(out.data_ptr()), st))
        if sync_errors:
            if _lib.lib().mdd_sync(self.handle, st) != 0:
                raise IndexError(_lib.lib().mdd_last_error().decode())
        return out

    def profile(self, x, x1):
        """Per-stage (name, ms, launches, flops) of one forward replayed stage by stage between HIP events."""
        B, T, _ = x.shape
        out = torch.empty((T // 2, B, self.geom.num_class), dtype=torch.float32, device=x.device)
        n = _lib.lib().mdd_forward_num_stages(self.handle)
        names = C.create_string_buffer(64 * n)
        ms, launches, flops = (C.c_float * n)(), (C.c_int32 * n)(), (C.c_double * n)()
        _lib.check(_lib.lib().mdd_forward_profile(self.handle, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(x1.data_ptr()),
                                                  x1.shape[1], C.c_void_p(out.data_ptr()), _lib.current_stream_ptr(),
                                                  names, 64 * n, ms, launches, flops, n))
        return list(zip(names.value.decode().split(","), list(ms), list(launches), list(flops)))

    def tap(self, name):
        n = C.c_int64(0)
        if not _lib.lib().mdd_tap(self.handle, name.encode(), C.byref(n)):
            raise KeyError(name)
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().mdd_tap_copy(self.handle, name.encode(), C.c_void_p(out.data_ptr()), n.value,
                                           _lib.current_stream_ptr()))
        return out

    def close(self):
        if self.handle:
            _lib.lib().mdd_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def ctc_loss(logp, targets, in_len, tgt_len, blank=0, want_grad=True):
    """nn.CTCLoss(reduction='sum') pieces on the GPU: returns (nll [B], grad [T,B,C] or None)."""
    assert logp.is_cuda and logp.dtype == torch.float32
    logp = logp.contiguous()
    T, B, Cn = logp.shape
    dev = logp.device
    tg = targets.to(dev, torch.int64).contiguous()
    il = in_len.to(dev, torch.int64).contiguous()
    tl = tgt_len.to(dev, torch.int64).contiguous()
    nll = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty_like(logp) if want_grad else None
    nws = _lib.lib().mdd_ctc_workspace_bytes(T, B, Cn, tg.shape[1], 1 if want_grad else 0)
    ws = torch.empty((max(nws, 16) + 7) // 8, dtype=torch.float64, device=dev)      # torch's caching allocator: stream-ordered, no hipMalloc per call
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mdd_ctc_loss(C.c_void_p(logp.data_ptr()), T, B, Cn, C.c_void_p(tg.data_ptr()), tg.shape[1],
                                           C.c_void_p(il.data_ptr()), C.c_void_p(tl.data_ptr()), blank,
                                           C.c_void_p(nll.data_ptr()), C.c_void_p(grad.data_ptr()) if want_grad else None,
                                           C.c_void_p(ws.data_ptr()), ws.numel() * 8, _lib.current_stream_ptr()))
    return nll, grad
