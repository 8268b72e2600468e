"""Batch construction at the boundary of the hot path -- mirrors ``utils/data_loader.py`` of the
reference (AA/utils/data_loader.py) for the parts the path needs: ``Vocab`` (units file -> ids,
blank = 0, UNK = 1; :13-52), the context-stack / frame-skip / even-pad of one utterance (:138-142,
done on the GPU by mdd_stack_skip) and the zero-padding collate with its float32 length fractions
(``create_input``, :151-181).  Reading Kaldi ark files is upstream of the path.
"""
import ctypes as C

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import _lib


class Vocab(object):
    def __init__(self, vocab_file):
        self.vocab_file = vocab_file
        self.word2index = {"blank": 0, "UNK": 1}
        self.index2word = {0: "blank", 1: "UNK"}
        self.word2count = {}
        self.n_words = 2
        self.read_lang()

    def add_word(self, word):
        if word in self.word2index:
            self.word2count[word] += 1
            return
        self.word2index[word] = self.n_words
        self.index2word[self.n_words] = word
        self.word2count[word] = 1
        self.n_words += 1

    def add_sentence(self, sentence):
        for word in sentence.split(" "):
            self.add_word(word)

    def read_lang(self):
        print("Reading vocabulary from {}".format(self.vocab_file))
        with open(self.vocab_file, "r") as rf:
            for line in rf:
                cols = line.strip().split(" ")
                self.add_sentence(" ".join(cols[1:]) if len(cols) > 1 else cols[0])
        print("Vocabulary size is {}".format(self.n_words))


def stack_features(raw, right_ctx=2, n_skip_frame=2, n_downsample=2, out=None):
    """raw [B, T_raw, D] (or [T_raw, D]) float tensor -> stacked [B, T, (right_ctx+1)*D] on the GPU:
    make_context(feat, 0, right_ctx) + skip_feat(., n_skip_frame) + zero-pad to a multiple of n_downsample."""
    _lib.require_gpu()
    single = raw.dim() == 2
    r = (raw.unsqueeze(0) if single else raw).to("cuda", torch.float32).contiguous()
    B, T_raw, D = r.shape
    T = _lib.lib().mdd_stack_len(T_raw, n_skip_frame, n_downsample)
    if out is None:
        out = torch.empty((B, T, (right_ctx + 1) * D), dtype=torch.float32, device=r.device)
    assert out.is_contiguous() and out.numel() == B * T * (right_ctx + 1) * D
    _lib.check(_lib.lib().mdd_stack_skip(C.c_void_p(r.data_ptr()), B, T_raw, D, right_ctx, n_skip_frame, n_downsample,
                                         C.c_void_p(out.data_ptr()), _lib.current_stream_ptr()))
    return out.view(T, -1) if single else out.view(B, T, -1)


def frames_from_fraction(input_sizes, t_out):
    """``(input_sizes * probs.size(0)).long()`` of AA/infer.py:296-297 -- float32 multiply, truncation."""
    return (input_sizes.float() * t_out).long()


def create_input(batch):
    """Collate of (feature [T,F], label [n], trans [L], utt) tuples: zero-pad everything to the batch
    maximum; lengths of the features as float32 fractions of the maximum."""
    t_max = max(item[0].size(0) for item in batch)
    n_max = max(item[1].size(0) for item in batch)
    l_max = max(item[2].size(0) for item in batch)
    B, F = len(batch), batch[0][0].size(1)
    data = torch.zeros(B, t_max, F)
    label = torch.zeros(B, n_max)
    trans = torch.zeros(B, l_max)
    in_sizes, lab_sizes, trans_sizes = torch.zeros(B), torch.zeros(B), torch.zeros(B)
    utts = []
    for i, (feat, lab, tr, utt) in enumerate(batch):
        data[i, :feat.size(0)] = feat
        label[i, :lab.size(0)] = lab
        trans[i, :tr.size(0)] = tr
        in_sizes[i] = feat.size(0) / t_max
        lab_sizes[i], trans_sizes[i] = lab.size(0), tr.size(0)
        utts.append(utt)
    return data.float(), in_sizes.float(), label.long(), lab_sizes.long(), trans.long(), trans_sizes.long(), utts


class SpeechDataset(Dataset):
    """Items of one data split: (stacked features [T, (ctx+1)*D], label ids, canonical-transcript ids, utterance id) --
    AA/utils/data_loader.py:54-146 for the fbank feature type.  ``scp_path``: Kaldi scp of ark offsets, ``lab_path`` /
    ``trans_path``: '<utt> <phoneme> <phoneme> ...' lines (annotated and canonical phonemes).  With ``train=True`` every
    item is augmented first (spec_augment on the raw frames, data_enhancement on the canonical ids; :132-137).  The
    stack / skip / even-pad runs on the host here (numpy, as in the reference's loader workers); batches that are already
    on the GPU use ``stack_features`` / ``mdd_forward_raw`` instead."""

    def __init__(self, vocab, scp_path, lab_path, trans_path, opts, train=False):
        self.vocab = vocab
        self.left_ctx, self.right_ctx = opts.left_ctx, opts.right_ctx
        self.n_skip_frame, self.n_downsample = opts.n_skip_frame, opts.n_downsample
        self.train = train
        unk = vocab.word2index["UNK"]

        def ids(path):
            table = {}
            with open(path, "r") as rf:
                for line in rf:
                    if line.strip():
                        utt, text = line.strip().split(" ", 1)
                        table[utt] = [vocab.word2index.get(c, unk) for c in text.split()]
            return table
        paths = []
        with open(scp_path, "r") as rf:
            for line in rf:
                if line.strip():
                    utt, path = line.strip().split(" ")
                    paths.append((utt.split(".")[0], path))
        labels, trans = ids(lab_path), ids(trans_path)
        assert len(paths) == len(labels) == len(trans)
        self.item = [(path, labels[utt], trans[utt], utt) for utt, path in paths]

    def __getitem__(self, idx):
        from .fbank import load_mat
        from .tools import augment_item, make_context, skip_feat
        path, label, trans, utt = self.item[idx]
        feat, trans = augment_item(load_mat(path), trans, train=self.train)
        feat = skip_feat(make_context(feat, self.left_ctx, self.right_ctx), self.n_skip_frame)
        if feat.shape[0] % self.n_downsample:
            feat = np.vstack([feat, np.zeros((self.n_downsample - feat.shape[0] % self.n_downsample, feat.shape[1]))])
        return torch.from_numpy(feat), torch.LongTensor(label), torch.LongTensor(trans), utt

    def __len__(self):
        return len(self.item)


class SpeechDataLoader(DataLoader):
    """DataLoader whose collate is ``create_input`` (AA/utils/data_loader.py:191-194)."""

    def __init__(self, *args, **kwargs):
        super(SpeechDataLoader, self).__init__(*args, **kwargs)
        self.collate_fn = create_input
