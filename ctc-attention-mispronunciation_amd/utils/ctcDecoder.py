"""Drop-in for the reference's ``utils/ctcDecoder.py`` (AA/utils/ctcDecoder.py).

``GreedyDecoder`` / ``BeamDecoder`` keep the reference's constructor arguments and
``decode(prob_tensor, frame_seq_len) -> list[str]`` contract (same strings, same exception types);
the scans run in gfx950 kernels (mdd_greedy / mdd_beam, csrc/decode.hip).  ``Decoder.wer`` keeps the
``(distance, op-path)`` return and runs the integer DP + backtrace natively on the host (mdd_align).
Posteriors may live on the GPU (no copy) or on the CPU as the reference passes them (copied in).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .NgramLM import LanguageModel

_OPS = ("-", "S", "I", "D")
_BEAM_ERRORS = {1: IndexError("tuple index out of range"), 2: ValueError("math domain error"), 3: KeyError}


class Decoder(object):
    """String conversion and error metrics shared by both decoders (ctcDecoder.py:9-184)."""

    def __init__(self, int2char, space_idx=1, blank_index=0):
        self.int_to_char = int2char
        self.space_idx = space_idx
        self.blank_index = blank_index
        self.num_word = 0
        self.num_char = 0

    def decode(self):
        raise NotImplementedError

    # -- string helpers (ctcDecoder.py:80-116)
    def _convert_to_string(self, seq, sizes):
        chars = [self.int_to_char[seq[i]] for i in range(sizes)]
        return chars if self.space_idx == -1 else "".join(chars)

    def _convert_to_strings(self, seq, sizes=None):
        return [self._convert_to_string(seq[x], sizes[x] if sizes is not None else len(seq[x])) for x in range(len(seq))]

    def _process_string(self, seq, remove_rep=False):
        out = ""
        blank = self.int_to_char[self.blank_index]
        for i, ch in enumerate(seq):
            if ch == blank or (remove_rep and i != 0 and ch == seq[i - 1]):
                continue
            if self.space_idx == -1:
                out += " " + ch
            elif ch == self.int_to_char[self.space_idx]:
                out += " "
            else:
                out += ch
        return out

    def _process_strings(self, seqs, remove_rep=False):
        return [self._process_string(s, remove_rep) for s in seqs]

    def _unflatten_targets(self, targets, target_sizes):
        out, off = [], 0
        for size in target_sizes:
            out.append(targets[off:off + size])
            off += size
        return out

    # -- metrics (ctcDecoder.py:118-184)
    def _edit_distance(self, src_seq, tgt_seq):
        """Reference quirk kept: a bare int (not a tuple) when either side is empty (ctcDecoder.py:137-138)."""
        if len(src_seq) == 0:
            return len(tgt_seq)
        if len(tgt_seq) == 0:
            return len(src_seq)
        dist, ops = align_ids(_tokens_to_ids(src_seq, tgt_seq))
        return dist, ops

    def wer(self, s1, s2):
        """Word-level edit distance of two space-separated strings and its op path
        ('-' match, 'S' substitution, 'I' token only in s1, 'D' token only in s2)."""
        ans, paths = self._edit_distance(s1.split(), s2.split())   # TypeError on an empty side, like the reference
        return ans, paths

    def cer(self, s1, s2):
        ans, _ = self._edit_distance(s1, s2)
        return ans

    def phone_word_error(self, prob_tensor, frame_seq_len, targets, target_sizes):
        strings = self.decode(prob_tensor, frame_seq_len)
        refs = self._process_strings(self._convert_to_strings(self._unflatten_targets(targets, target_sizes)))
        cer = wer = 0
        for hyp, ref in zip(strings, refs):
            cer += self.cer(hyp, ref)
            wer += self.wer(hyp, ref)[0]
            self.num_word += len(ref.split())
            self.num_char += len(ref)
        return cer, wer


def _tokens_to_ids(a, b):
    table = {}
    ia = [table.setdefault(t, len(table)) for t in a]
    ib = [table.setdefault(t, len(table)) for t in b]
    return ia, ib


def align_ids(pair):
    """(dist, ops) for two non-empty integer sequences via the native mdd_align."""
    a = np.ascontiguousarray(pair[0], dtype=np.int32)
    b = np.ascontiguousarray(pair[1], dtype=np.int32)
    ops = np.zeros(len(a) + len(b) + 1, dtype=np.uint8)
    dist, nops = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().mdd_align(a.ctypes.data_as(C.c_void_p), len(a), b.ctypes.data_as(C.c_void_p), len(b),
                                    C.byref(dist), ops.ctypes.data_as(C.c_void_p), C.byref(nops)))
    return dist.value, [_OPS[o] for o in ops[:nops.value]]


def align_ids_batch(a, a_len, b, b_len):
    """Edit distance and op path of every row pair of two padded id matrices in one native call (mdd_align_batch).

    a [n, Sa], b [n, Sb] int32 with per-row lengths; returns (dist [n], ops [n, Sa+Sb] uint8 codes 0 '-', 1 'S', 2 'I',
    3 'D', nops [n]).  A row with an empty side has dist -1 and no ops (the reference's ``wer`` raises there)."""
    a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
    a_len = np.ascontiguousarray(a_len, dtype=np.int32); b_len = np.ascontiguousarray(b_len, dtype=np.int32)
    if a.ndim != 2 or b.ndim != 2 or a.shape[0] != b.shape[0] or a_len.shape != (a.shape[0],) or b_len.shape != (a.shape[0],):
        raise ValueError("align_ids_batch: a [n, Sa], b [n, Sb], a_len [n], b_len [n]")
    n, stride = a.shape[0], a.shape[1] + b.shape[1]
    dist, nops = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    ops = np.zeros((n, max(stride, 1)), dtype=np.uint8)
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.lib().mdd_align_batch(p(a), p(a_len), a.shape[1], p(b), p(b_len), b.shape[1], n, p(dist), p(ops), ops.shape[1], p(nops)))
    return dist, ops, nops


def _device_posteriors(prob_tensor):
    _lib.require_gpu()
    if prob_tensor.dim() != 3:
        raise ValueError("prob_tensor must be [T, B, C]")
    dev = prob_tensor.device if prob_tensor.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return prob_tensor.to(dev, torch.float32).contiguous()


def _lens_tensor(frame_seq_len, B, T, dev):
    if frame_seq_len is None:
        frame_seq_len = [T] * B
    if torch.is_tensor(frame_seq_len):
        return frame_seq_len.to(dev, torch.int32).contiguous()
    return torch.tensor([int(v) for v in frame_seq_len], dtype=torch.int32, device=dev)


class GreedyDecoder(Decoder):
    """argmax per frame, collapse repeats, drop blanks (ctcDecoder.py:186-200)."""

    def decode_ids(self, prob_tensor, frame_seq_len=None):
        lp = _device_posteriors(prob_tensor)
        T, B, Cn = lp.shape
        lens = _lens_tensor(frame_seq_len, B, T, lp.device)
        ids = torch.empty((B, T), dtype=torch.int32, device=lp.device)
        nids = torch.empty((B,), dtype=torch.int32, device=lp.device)
        with torch.cuda.device(lp.device):
            _lib.check(_lib.lib().mdd_greedy(C.c_void_p(lp.data_ptr()), T, B, Cn, C.c_void_p(lens.data_ptr()),
                                             self.blank_index, C.c_void_p(ids.data_ptr()), C.c_void_p(nids.data_ptr()),
                                             _lib.current_stream_ptr()))
        return ids, nids

    def decode(self, prob_tensor, frame_seq_len):
        ids, nids = self.decode_ids(prob_tensor, frame_seq_len)
        ids, nids = ids.cpu().numpy(), nids.cpu().numpy()
        out = []
        for b in range(ids.shape[0]):
            chars = [self.int_to_char[int(i)] for i in ids[b, :nids[b]]]
            if self.space_idx == -1:
                out.append("".join(" " + c for c in chars))
            else:
                sp = self.int_to_char[self.space_idx]
                out.append("".join(" " if c == sp else c for c in chars))
        return out


class BeamDecoder(Decoder):
    """CTC prefix beam search with a bigram LM hook (ctcDecoder.py:202-226, BeamSearch.py:73-153)."""

    def __init__(self, int2char, beam_width=200, blank_index=0, space_idx=-1, lm_path=None, lm_alpha=0.01):
        self.beam_width = beam_width
        super(BeamDecoder, self).__init__(int2char, space_idx=space_idx, blank_index=blank_index)
        self.lm = LanguageModel(arpa_file=lm_path)
        self.lm_alpha = lm_alpha
        self._tables = {}

    def _lm_table(self, num_class, dev):
        key = (num_class, str(dev))
        if key not in self._tables:
            self._tables[key] = torch.from_numpy(self.lm.dense_table(self.int_to_char, num_class)).to(dev)
        return self._tables[key]

    def decode_ids(self, prob_tensor, frame_seq_len=None):
        lp = _device_posteriors(prob_tensor)
        T, B, Cn = lp.shape
        lens = _lens_tensor(frame_seq_len, B, T, lp.device)
        ids = torch.empty((B, T), dtype=torch.int32, device=lp.device)
        nids = torch.empty((B,), dtype=torch.int32, device=lp.device)
        status = torch.empty((B,), dtype=torch.int32, device=lp.device)
        score = torch.empty((B,), dtype=torch.float64, device=lp.device)
        lm = self._lm_table(Cn, lp.device)
        with torch.cuda.device(lp.device):
            _lib.check(_lib.lib().mdd_beam(C.c_void_p(lp.data_ptr()), T, B, Cn, C.c_void_p(lens.data_ptr()), self.beam_width,
                                           self.blank_index, C.c_void_p(lm.data_ptr()), float(self.lm_alpha),
                                           C.c_void_p(ids.data_ptr()), C.c_void_p(nids.data_ptr()),
                                           C.c_void_p(status.data_ptr()), C.c_void_p(score.data_ptr()),
                                           _lib.current_stream_ptr()))
        return ids, nids, status, score

    def decode(self, prob_tensor, frame_seq_len=None):
        ids, nids, status, _ = self.decode_ids(prob_tensor, frame_seq_len)
        ids, nids, status = ids.cpu().numpy(), nids.cpu().numpy(), status.cpu().numpy()
        bad = np.nonzero(status)[0]
        if len(bad):   # the reference stops at the first utterance that raises
            err = _BEAM_ERRORS[int(status[bad[0]])]
            raise err("utterance %d of the batch" % bad[0]) if err is KeyError else err
        return [" ".join(self.int_to_char[int(i)] for i in ids[b, :nids[b]]) for b in range(ids.shape[0])]
