"""Batch evaluation of a decoded batch: the step right after the hot path.

Mirror of the per-batch bookkeeping inside ``test()`` of the reference's ``AA/steps/test_ctc_nosil.py``:

* ``print_align_space_canonical_origin(s1, s2, l)``  -- same name, arguments and 4-tuple return (:33-82);
* ``strip_sil`` -- the "compute with out sil" loop (:196-209);
* ``count_batch`` -- the TA / FR / FA / TR tallies, the canonical phoneme count and the PER numerator / denominator of
  one batch (:218-298), computed natively by ``mdd_eval_batch`` (C ABI, host) on integer ids;
* ``MddCounts.report()`` -- precision / recall / F1 / PER exactly as printed at :300-318 (``ZeroDivisionError``
  included when a denominator is zero).

The reference keeps these numbers in local variables of one long function; here they are a small accumulator class so
that ranks can sum them (``MddCounts.__add__``) after an all-gather of eight integers.
"""
import ctypes as C

import numpy as np

from .. import _lib


def print_align_space_canonical_origin(s1, s2, l):
    """(canonical line, hypothesis line, op line, d) for hypothesis ``s1``, canonical ``s2`` and their op path ``l``.

    ``d[j]`` is what happened to canonical position j: '-', 'D', or 'S' + the hypothesis phoneme; ``d['I']`` lists the
    insertions as the string str(j-1) + str(j) of the canonical gap they fall in.  The two text lines carry a 'D' / 'I'
    placeholder where the other side has no token, every token padded to three characters.
    """
    hyp, can = s1.split(' '), s2.split(' ')
    d = {j: "" for j in range(len(can))}
    d['I'] = []
    hyp_line, can_line = [], []
    hi = ci = 0
    for op in l:
        if op == '-' or op == 'S':
            d[ci] = op + (hyp[hi] if op == 'S' else "")
            hyp_line.append(hyp[hi]); can_line.append(can[ci])
            hi += 1; ci += 1
        elif op == 'D':
            d[ci] = 'D'
            hyp_line.append('D'); can_line.append(can[ci])
            ci += 1
        else:
            d['I'].append(str(ci - 1) + str(ci))
            hyp_line.append(hyp[hi]); can_line.append('I')
            hi += 1
    hyp_line += hyp[hi:]                 # a path shorter than the strings leaves the tails as they are
    can_line += can[ci:]

    def pad(tok):
        return tok + " " * max(0, 3 - len(tok)) if len(tok) in (1, 2) else tok

    return (' '.join(pad(t) for t in can_line), ' '.join(pad(t) for t in hyp_line),
            ' '.join(s + "  " for s in l), d)


def strip_sil(strings):
    """Drop the 'sil' tokens of space-separated strings (split on single spaces, like the reference)."""
    return [' '.join(t for t in s.split(" ") if t != "sil") for s in strings]


class MddCounts(object):
    """Running totals of the evaluation: canonical phonemes, TA, FR, FA, TR (correct / wrong diagnosis), PER terms."""
    FIELDS = ("total", "TA", "FR", "FA", "TR_correct", "TR_wrong", "total_error", "total_phoneme")

    def __init__(self, values=None):
        vals = [0] * 8 if values is None else [int(v) for v in values]
        for k, v in zip(self.FIELDS, vals):
            setattr(self, k, v)

    def as_list(self):
        return [getattr(self, k) for k in self.FIELDS]

    def __add__(self, other):
        return MddCounts([a + b for a, b in zip(self.as_list(), other.as_list())])

    @property
    def TR(self):
        return self.TR_correct + self.TR_wrong

    def report(self):
        """dict(PER, precision, recall, F1) in percent; ZeroDivisionError where the reference would raise it."""
        per = float(self.total_error) / self.total_phoneme * 100
        p = float(self.TR) / (self.TR + self.FR)
        r = float(self.TR) / (self.TR + self.FA)
        return dict(PER=per, precision=p * 100, recall=r * 100, F1=2 * p * r / (p + r) * 100)


def count_batch(decoded, labels, canonicals, remove_sil=True):
    """MddCounts of one batch of space-separated phoneme strings (decoded, annotated, canonical).

    Raises TypeError when a sequence is empty after 'sil' removal, as the reference's ``decoder.wer`` does.
    """
    if not (len(decoded) == len(labels) == len(canonicals)):
        raise ValueError("decoded, labels and canonicals must have the same length")
    if remove_sil:
        decoded, labels, canonicals = strip_sil(decoded), strip_sil(labels), strip_sil(canonicals)
    ids = {}
    seqs = [[[ids.setdefault(t, len(ids)) for t in s.split()] for s in group] for group in (decoded, labels, canonicals)]
    n = len(decoded)
    stride = max([1] + [len(v) for group in seqs for v in group])
    arr = np.zeros((3, max(n, 1), stride), dtype=np.int32)
    lens = np.zeros((3, max(n, 1)), dtype=np.int32)
    for g in range(3):
        for x, v in enumerate(seqs[g]):
            arr[g, x, :len(v)] = v
            lens[g, x] = len(v)
    counts = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = _lib.lib().mdd_eval_batch(p(arr[0]), p(lens[0]), p(arr[1]), p(lens[1]), p(arr[2]), p(lens[2]), n, stride, p(counts))
    if rc == _lib.MDD_ERR_EMPTY:
        raise TypeError("cannot unpack non-iterable int object")      # what `_, path = decoder.wer(...)` raises in the reference
    _lib.check(rc)
    return MddCounts(counts)
