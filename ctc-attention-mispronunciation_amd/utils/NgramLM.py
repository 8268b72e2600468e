"""ARPA bigram language model reader -- drop-in for the reference's ``utils/NgramLM.py``.

Reference: AA/utils/NgramLM.py:25-78.  Same class name, constructor and ``get_bi_prob`` contract
(natural-log scores; back-off ``unigram[w1].backoff + unigram[w2].prob`` when the bigram is absent;
KeyError when a needed unigram is absent).  ``dense_table`` flattens it into the (C+1)x(C+1) array
the GPU beam search consumes (include/mdd_hip.h, mdd_beam).
"""
import math

import numpy as np


class LanguageModel(object):
    def __init__(self, arpa_file=None, n_gram=2, start="<s>", end="</s>", unk="<unk>"):
        self.n_gram, self.start, self.end, self.unk = n_gram, start, end, unk
        self.scale = math.log(10)   # ARPA stores log10
        self.unigram, self.bigram = {}, {}
        self._read(arpa_file)

    def _read(self, path):
        section = 0
        with open(path, "r") as f:
            for raw in f:
                line = raw.rstrip("\n")
                if line == "\\1-grams:":
                    section = 1
                elif line == "\\2-grams:":
                    section = 2
                elif section:
                    cols = line.split("\t")
                    if len(cols) in (2, 3):
                        table = self.unigram if section == 1 else self.bigram
                        backoff = self.scale * float(cols[2]) if len(cols) == 3 else 0.0
                        table[cols[1]] = [self.scale * float(cols[0]), backoff]
        self.unigram["UNK"] = self.unigram[self.unk]

    def get_uni_prob(self, wid):
        return self.unigram[wid][0]

    def get_bi_prob(self, w1, w2):
        w1 = w1 if w1 != "" else self.start
        w2 = w2 if w2 != "" else self.end
        hit = self.bigram.get(w1 + " " + w2)
        if hit is not None:
            return hit[0]
        return self.unigram[w1][1] + self.unigram[w2][0]

    def score_bg(self, sentence):
        words = sentence.strip().split()
        seq = [self.start] + words + [self.end]
        return sum(self.get_bi_prob(a, b) for a, b in zip(seq[:-1], seq[1:]))

    def dense_table(self, int2char, num_class):
        """T[prev][next] float64, prev/next == num_class meaning sentence start / end; NaN where
        ``get_bi_prob`` would raise KeyError (the beam kernel turns that into the same exception)."""
        tab = np.full((num_class + 1, num_class + 1), np.nan, dtype=np.float64)
        names = [int2char[i] for i in range(num_class)] + [""]
        for p, c1 in enumerate(names):
            for n, c2 in enumerate(names):
                try:
                    tab[p, n] = self.get_bi_prob(c1, c2)
                except KeyError:
                    pass
        return tab
