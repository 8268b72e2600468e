"""Drop-in for the reference's ``models/model_ctc.py`` (AA/models/model_ctc.py).

Same class names, constructor arguments, ``state_dict`` keys, ``forward(x, x1[, visualize])``
signature and return layout ([T/2, B, C] log-probabilities) as the reference, so that
``from models.model_ctc import *`` in AA/infer.py:24 can be pointed here unchanged.  The torch
modules below are parameter containers only (they give the 61 reference key names and make
``load_state_dict`` / ``.to()`` work); the arithmetic of ``forward`` runs in hand-written gfx950
kernels behind the C ABI of libmdd_hip.so (include/mdd_hip.h): eval mode through mdd_forward, train mode (BatchNorm on batch
statistics, dropout, differentiable w.r.t. every parameter) through mdd_train_forward / mdd_train_backward (train.py).

AA/infer.py also relies on this module re-exporting ``math`` (infer.py:342) -- kept, together with
the other names the reference module exposes through ``import *``.
"""
import ctypes as C
import math  # noqa: F401  (re-exported on purpose)
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

from .. import _lib


class _EditDistance(object):
    """Stand-in for the third-party ``editdistance`` module the reference imports as ``ed``
    (model_ctc.py:7); only ``ed.eval`` is used (model_ctc.py:240)."""

    @staticmethod
    def eval(a, b):
        a, b = list(a), list(b)
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            row = [i]
            for j, cb in enumerate(b, 1):
                row.append(min(row[j - 1] + 1, prev[j] + 1, prev[j - 1] + (ca != cb)))
            prev = row
        return prev[len(b)]


ed = _EditDistance()


class BatchRNN(nn.Module):
    """Parameter container with the reference's names: ``batch_norm.*`` (layers > 0) and
    ``rnn.weight_{ih,hh}_l0[_reverse]`` (model_ctc.py:15-34)."""

    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM, bidirectional=False, batch_norm=True,
                 dropout=0.1, skip=False):
        super(BatchRNN, self).__init__()
        if skip:
            raise NotImplementedError("skip projections are never enabled by the reference recipe")
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        self.batch_norm = nn.BatchNorm1d(input_size) if batch_norm else None
        self.rnn = rnn_type(input_size=input_size, hidden_size=hidden_size, bidirectional=bidirectional, bias=False)


class LayerCNN(nn.Module):
    """Parameter container: ``conv.*`` + ``batch_norm.*`` (model_ctc.py:51-71)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, padding, pooling_size=None,
                 activation_function=nn.ReLU, batch_norm=True, dropout=0.1):
        super(LayerCNN, self).__init__()
        self.conv = nn.Conv2d(in_channel, out_channel, kernel_size=kernel_size, stride=stride, padding=padding)
        self.batch_norm = nn.BatchNorm2d(out_channel) if batch_norm else None
        self.geometry = (tuple(kernel_size), tuple(stride), tuple(padding), pooling_size)


class CTC_Model(nn.Module):
    def __init__(self, add_cnn=False, cnn_param=None, rnn_param=None, num_class=39, drop_out=0.1):
        super(CTC_Model, self).__init__()
        self.add_cnn = add_cnn
        self.cnn_param = cnn_param
        if rnn_param is None or type(rnn_param) != dict:
            raise ValueError("rnn_param need to be a dict to contain all params of rnn!")
        self.rnn_param = rnn_param
        self.num_class = num_class
        self.num_directions = 2 if rnn_param["bidirectional"] else 1
        self.drop_out = drop_out
        feat = rnn_param["rnn_input_size"]
        width = feat
        if add_cnn:
            layers = []
            for n, (chan, ksz, stride, pad, pool) in enumerate(cnn_param["layer"]):
                layers.append(("%d" % n, LayerCNN(chan[0], chan[1], ksz, stride, pad, pool,
                                                  activation_function=cnn_param["activate_function"],
                                                  batch_norm=cnn_param["batch_norm"], dropout=drop_out)))
                width = int(math.floor((width + 2 * pad[1] - ksz[1]) / stride[1]) + 1)
            self.conv = nn.Sequential(OrderedDict(layers))
            width *= cnn_param["layer"][-1][0][1]
        H = rnn_param["rnn_hidden_size"]
        rnns = [("0", BatchRNN(width, H, rnn_type=rnn_param["rnn_type"], bidirectional=rnn_param["bidirectional"],
                               dropout=drop_out, batch_norm=False))]
        for i in range(rnn_param["rnn_layers"] - 1):
            rnns.append(("%d" % (i + 1), BatchRNN(self.num_directions * H, H, rnn_type=rnn_param["rnn_type"],
                                                  bidirectional=rnn_param["bidirectional"], dropout=drop_out,
                                                  batch_norm=rnn_param["batch_norm"])))
        self.rnns = nn.Sequential(OrderedDict(rnns))
        self.embeds = nn.Embedding(44, 512)                                       # model_ctc.py:149
        self.lstm_embeds = nn.LSTM(512, H, batch_first=True, bidirectional=True)  # :150
        self.score = nn.Linear(H * 2, H * 2, bias=False)                          # :151
        if rnn_param["batch_norm"]:
            self.fc = nn.Sequential(nn.BatchNorm1d(self.num_directions * H * 2),
                                    nn.Linear(self.num_directions * H * 2, num_class, bias=False))
        else:
            self.fc = nn.Linear(self.num_directions * H * 2, num_class, bias=False)
        self.log_softmax = nn.LogSoftmax(dim=-1)
        self._handle = None
        self._dirty = True
        self.strict_errors = True   # raise IndexError for bad canonical ids synchronously, like nn.Embedding
        self._dropout_masks = None  # tests: the reference's own dropout masks (one uint8 tensor per site, mdd_hip.h); None = drawn
        self.train_precision = None  # train mode: None = library default (exact fp32, or MDD_TRAIN_PRECISION), "f32", or "bf16x3" (flagged variant)

    # ------------------------------------------------------------------ library plumbing
    def _check_supported(self):
        p, c = self.rnn_param, self.cnn_param
        ok = (self.add_cnn and p["bidirectional"] and p["batch_norm"] and c["batch_norm"] and p["rnn_type"] is nn.LSTM
              and len(c["layer"]) == 2
              and [l.geometry for l in self.conv] == [((3, 3), (1, 2), (1, 1), None), ((3, 3), (2, 2), (1, 1), None)]
              and c["layer"][0][0][0] == 1 and c["layer"][0][0][1] == c["layer"][1][0][0] == c["layer"][1][0][1])
        if not ok:
            raise NotImplementedError("libmdd_hip implements the architecture of conf/ctc_config.*.yaml "
                                      "(2x conv k3 s(1,2),(2,2) p1 + BN, BiLSTM + BN); got something else")

    def _config(self):
        return _lib.MddConfig(feat=self.rnn_param["rnn_input_size"], hidden=self.rnn_param["rnn_hidden_size"],
                              layers=self.rnn_param["rnn_layers"], num_class=self.num_class,
                              channels=self.cnn_param["layer"][0][0][1], emb_rows=self.embeds.num_embeddings,
                              emb_dim=self.embeds.embedding_dim, bn_eps=self.fc[0].eps)

    def _sync_weights(self, device_index):
        L = _lib.lib()
        if self._handle is None:
            self._check_supported()
            h = C.c_void_p()
            cfg = self._config()
            _lib.check(L.mdd_create(C.byref(cfg), device_index, C.byref(h)))
            self._handle = h
            self._dirty = True
        if not self._dirty:
            return
        for key, t in self.state_dict().items():
            if not t.is_floating_point():
                continue
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(L.mdd_load_weight(self._handle, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        _lib.check(L.mdd_finalize_weights(self._handle))
        self._dirty = False

    def load_state_dict(self, *args, **kwargs):
        self._dirty = True
        return super(CTC_Model, self).load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._dirty = True
        return super(CTC_Model, self)._apply(fn, *args, **kwargs)

    def mark_weights_changed(self):
        """Call after editing parameters in place."""
        self._dirty = True

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().mdd_destroy(self._handle)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def enable_taps(self, on=True):
        self._taps = bool(on)
        if self._handle is not None:
            _lib.check(_lib.lib().mdd_enable_taps(self._handle, int(on)))

    def tap(self, name):
        """Stage output of the last forward as a torch tensor (copy): conv1, rnn<i>, text, key."""
        n = C.c_int64(0)
        if not _lib.lib().mdd_tap(self._handle, name.encode(), C.byref(n)):
            raise KeyError(name)
        out = torch.empty(n.value, dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().mdd_tap_copy(self._handle, name.encode(), C.c_void_p(out.data_ptr()), n.value,
                                           _lib.current_stream_ptr()))
        return out

    # ------------------------------------------------------------------ the reference API
    def forward(self, x, x1, visualize=False):
        """x: [B, T, F] float stacked features (T even); x1: [B, L] long canonical phoneme ids.
        Returns log-probabilities [T/2, B, num_class] on x's device (model_ctc.py:160-223)."""
        if not self.add_cnn:
            print("error")          # model_ctc.py:224-225
            return None
        _lib.require_gpu()
        if self.training:      # BatchNorm on batch statistics, Dropout(drop_out) behind each LayerCNN / BatchRNN; differentiable
            if visualize:
                raise NotImplementedError("visualize=True is an eval-mode facility here")
            from ..train import model_forward_train
            return model_forward_train(self, x, x1, masks=getattr(self, "_dropout_masks", None))
        src_device = x.device
        dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        xd = x.to(dev, torch.float32).contiguous()
        idd = x1.to(dev, torch.int64).contiguous()
        B, T, Fdim = xd.shape
        if Fdim != self.rnn_param["rnn_input_size"]:
            raise RuntimeError("expected feature size %d, got %d" % (self.rnn_param["rnn_input_size"], Fdim))
        with torch.cuda.device(dev):
            self._sync_weights(dev.index if dev.index is not None else torch.cuda.current_device())
            if getattr(self, "_taps", False) or visualize:
                _lib.check(_lib.lib().mdd_enable_taps(self._handle, 1))
            out = torch.empty((T // 2, B, self.num_class), dtype=torch.float32, device=dev)
            st = _lib.current_stream_ptr()
            _lib.check(_lib.lib().mdd_forward(self._handle, C.c_void_p(xd.data_ptr()), B, T, C.c_void_p(idd.data_ptr()),
                                              idd.shape[1], C.c_void_p(out.data_ptr()), st))
            if self.strict_errors:
                rc = _lib.lib().mdd_sync(self._handle, st)
                if rc != 0:
                    raise IndexError(_lib.lib().mdd_last_error().decode())
            if visualize:
                ch = self.cnn_param["layer"][-1][0][1]
                seq = self.tap("conv1").view(T // 2, B, -1)
                cnn = seq.view(T // 2, B, ch, -1).permute(1, 2, 0, 3).contiguous()
                res = out.to(src_device)
                return res, [x, cnn.to(src_device), seq.to(src_device), res]
        return out if src_device == dev else out.to(src_device)

    def compute_wer(self, index, input_sizes, targets, target_sizes):
        """Greedy collapse + edit distance on host index arrays (model_ctc.py:227-244)."""
        errs = toks = 0
        for i in range(len(index)):
            label = targets[i][:target_sizes[i]]
            frames = index[i][:input_sizes[i]]
            pred = [frames[j] for j in range(len(frames))
                    if frames[j] != 0 and (j == 0 or frames[j] != frames[j - 1])]
            errs += ed.eval(label, pred)
            toks += len(label)
        return errs, toks

    def add_weights_noise(self):
        """model_ctc.py:246-249: draws N(0, 0.075) noise per parameter and binds the sum to a LOCAL name -- the parameters are left
        as they are; what the call does observably is advance the generator, which is kept."""
        for param in self.parameters():
            param.data.new(param.size()).normal_(0, 0.075)

    @staticmethod
    def save_package(model, optimizer=None, decoder=None, epoch=None, loss_results=None, dev_loss_results=None,
                     dev_cer_results=None):
        """Checkpoint dict with the reference's keys (model_ctc.py:251-271)."""
        package = {"rnn_param": model.rnn_param, "add_cnn": model.add_cnn, "cnn_param": model.cnn_param,
                   "num_class": model.num_class, "_drop_out": model.drop_out, "state_dict": model.state_dict()}
        for key, val in (("optim_dict", optimizer.state_dict() if optimizer is not None else None),
                         ("decoder", decoder), ("epoch", epoch)):
            if val is not None:
                package[key] = val
        if loss_results is not None:
            package.update(loss_results=loss_results, dev_loss_results=dev_loss_results, dev_cer_results=dev_cer_results)
        return package
