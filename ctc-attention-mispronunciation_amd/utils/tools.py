"""Host-side feature / transcript helpers at the boundary of the hot path -- mirrors the parts of the reference's
``utils/tools.py`` (AA/utils/tools.py) that the data loader applies to every training item:

* ``make_context`` / ``skip_feat`` (:207-227) -- numpy forms of the stack/skip that ``mdd_stack_skip`` and the fused conv
  front-end perform on the GPU (kept for host-side pipelines and as the shape contract);
* ``spec_augment`` (:229-255) -- one frequency mask and one time mask per call, widths drawn from numpy's global RNG and
  positions from Python's ``random`` (both generators are consumed in the reference's order, so a seeded run reproduces
  the reference's masks exactly);
* ``data_enhancement`` (:290-359) -- canonical-phoneme mutation applied to the attention branch's transcript
  (AA/utils/data_loader.py:132-137): type 1 vowel->vowel / consonant->consonant, type 2 confusion-table draw, type 3
  uniform shift, type 4 zeroing.  ``random.choice`` indexes the lists below, so their ORDER is part of the contract.

The 41-phone class table (``word2index`` / ``index2word``, :58-104) is ``synth.phone_table_41``.
"""
import json
import os
import random

import numpy as np

from ..synth import phone_table_41

index2word = phone_table_41()
word2index = {w: i for i, w in index2word.items()}

# order matters: random.choice(seq) returns seq[floor(random() * len(seq))]  (AA/utils/tools.py:55-56)
vowels = "iy aa ae eh ah ao ih ey aw ay er uw uh oy ow ah0 er0".split()
consonants = "w dh y hh ch jh th zh d ng b g f k m l n s r t v z p sh".split()

_CONFUSIONS = None


def common_incorrect_voc():
    """The reference's phoneme confusion lists (AA/utils/tools.py:12-53), shipped as a data file next to this module
    (written by oracle/gen_golden.py from the reference's table)."""
    global _CONFUSIONS
    if _CONFUSIONS is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "common_incorrect_voc.json")) as f:
            _CONFUSIONS = json.load(f)
    return _CONFUSIONS


def make_context(feature, left, right):
    """Concatenate each frame with `left` previous and `right` following frames (edge frames replicated)."""
    feature = np.asarray(feature)
    T = feature.shape[0]
    cols = [feature]
    for k in range(1, left + 1):                       # frame t-k, the first frame repeated in front
        cols.append(feature[np.maximum(np.arange(T) - k, 0)])
    for k in range(1, right + 1):                      # frame t+k, the last frame repeated behind
        cols.append(feature[np.minimum(np.arange(T) + k, T - 1)])
    return np.hstack(cols)


def skip_feat(feature, skip):
    """Keep every `skip`-th frame, starting with frame 0."""
    feature = np.asarray(feature)
    return feature if skip in (0, 1) else feature[::skip]


def spec_augment(mel_spectrogram, frequency_mask_num=1, time_mask_num=1, frequency_masking_para=2, time_masking_para=5):
    tau, v = mel_spectrogram.shape[0], mel_spectrogram.shape[1]
    out = np.array(mel_spectrogram)
    for _ in range(max(frequency_mask_num, 0)):
        f = int(np.random.uniform(low=0.0, high=frequency_masking_para))
        f0 = random.randint(0, v - f)
        out[:, f0:f0 + f] = 0
    for _ in range(max(time_mask_num, 0)):
        t = int(np.random.uniform(low=0.0, high=time_masking_para))
        t0 = random.randint(0, tau - t)
        out[t0:t0 + t, :] = 0
    return out


def data_enhancement(phone, mutation_prob=0.1, enhancement_type=1, phone_num=44):
    """One canonical phoneme id -> a one-element list holding the (possibly mutated) id."""
    out = phone
    if enhancement_type == 1:
        if random.random() < mutation_prob:
            name = index2word[out]
            if name in vowels:
                out = word2index[random.choice(vowels)]
            elif name in consonants:
                out = word2index[random.choice(consonants)]
    elif enhancement_type == 2:
        if random.random() < mutation_prob:
            table = common_incorrect_voc()
            name = index2word[out]
            if name in table:
                out = word2index[random.choice(table[name])]
    elif enhancement_type == 3:
        if random.random() < mutation_prob:
            out = (phone + random.randint(0, phone_num)) % phone_num
    elif enhancement_type == 4:
        if random.random() < mutation_prob:
            out = 0
    return [out]


def augment_item(feat, trans, train=True):
    """What SpeechDataset.__getitem__ does to a training item before stacking (AA/utils/data_loader.py:132-137)."""
    if not train:
        return feat, list(trans)
    feat = spec_augment(feat)
    return feat, sum([data_enhancement(t) for t in trans], [])
