"""Deterministic synthetic weights and inputs for the CTC-attention MDD hot path.

There is no trained checkpoint in the reference tree, so every
parity test, golden vector and bench run uses a state_dict regenerated from a
seed on both sides.  Key names and shapes are exactly the 61 entries of the
reference ``CTC_Model.state_dict()`` (AA/models/model_ctc.py:84-158).

Pure numpy (PCG64) -- no torch RNG, so the stream is stable across torch builds.
"""
from collections import OrderedDict

import numpy as np


class Geometry(object):
    """Shapes of one model instance (reference: AA/conf/ctc_config.0329.yaml:50-64)."""

    def __init__(self, feat=243, hidden=384, layers=4, num_class=45,
                 channels=32, emb_rows=44, emb_dim=512):
        self.feat = feat            # stacked input width (81 * 3)
        self.hidden = hidden        # rnn_hidden_size
        self.layers = layers        # rnn_layers
        self.num_class = num_class  # blank + UNK + units
        self.channels = channels    # conv channels (both layers)
        self.emb_rows = emb_rows    # nn.Embedding(44, 512), model_ctc.py:149
        self.emb_dim = emb_dim

    @property
    def w1(self):  # width after conv0: k3, stride 2, pad 1
        return (self.feat + 2 - 3) // 2 + 1

    @property
    def w2(self):  # width after conv1
        return (self.w1 + 2 - 3) // 2 + 1

    @property
    def rnn_in(self):
        return self.channels * self.w2

    def cnn_param(self, nn):
        c = self.channels
        return {"layer": [[(1, c), (3, 3), (1, 2), (1, 1), None],
                          [(c, c), (3, 3), (2, 2), (1, 1), None]],
                "batch_norm": True, "activate_function": nn.ReLU}

    def rnn_param(self, nn):
        return {"rnn_input_size": self.feat, "rnn_hidden_size": self.hidden,
                "rnn_layers": self.layers, "rnn_type": nn.LSTM,
                "bidirectional": True, "batch_norm": True}


REFERENCE = dict(feat=243, hidden=384, layers=4, num_class=45)
REFERENCE_256 = dict(feat=243, hidden=256, layers=4, num_class=45)
TINY = dict(feat=15, hidden=8, layers=2, num_class=7, channels=4, emb_rows=7, emb_dim=12)


def _uni(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _bn(rng, sd, prefix, n):
    sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    sd[prefix + ".bias"] = rng.uniform(-0.3, 0.3, size=n).astype(np.float32)
    sd[prefix + ".running_mean"] = (0.2 * rng.standard_normal(n)).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.array(7, dtype=np.int64)


def synth_state_dict(geom, seed=1234, fc_gain=6.0):
    """Ordered dict key -> numpy array, keys/shapes as the reference state_dict."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    c, H = geom.channels, geom.hidden
    sd["conv.0.conv.weight"] = _uni(rng, (c, 1, 3, 3), 1.0 / 3.0)
    sd["conv.0.conv.bias"] = _uni(rng, (c,), 1.0 / 3.0)
    _bn(rng, sd, "conv.0.batch_norm", c)
    b = 1.0 / np.sqrt(9.0 * c)
    sd["conv.1.conv.weight"] = _uni(rng, (c, c, 3, 3), b)
    sd["conv.1.conv.bias"] = _uni(rng, (c,), b)
    _bn(rng, sd, "conv.1.batch_norm", c)
    k = 1.0 / np.sqrt(H)
    for n in range(geom.layers):
        isz = geom.rnn_in if n == 0 else 2 * H
        if n > 0:
            _bn(rng, sd, "rnns.%d.batch_norm" % n, 2 * H)
        for sfx in ("", "_reverse"):
            sd["rnns.%d.rnn.weight_ih_l0%s" % (n, sfx)] = _uni(rng, (4 * H, isz), k)
            sd["rnns.%d.rnn.weight_hh_l0%s" % (n, sfx)] = _uni(rng, (4 * H, H), k)
    sd["embeds.weight"] = rng.standard_normal((geom.emb_rows, geom.emb_dim)).astype(np.float32)
    for sfx in ("", "_reverse"):
        sd["lstm_embeds.weight_ih_l0" + sfx] = _uni(rng, (4 * H, geom.emb_dim), k)
        sd["lstm_embeds.weight_hh_l0" + sfx] = _uni(rng, (4 * H, H), k)
        sd["lstm_embeds.bias_ih_l0" + sfx] = _uni(rng, (4 * H,), k)
        sd["lstm_embeds.bias_hh_l0" + sfx] = _uni(rng, (4 * H,), k)
    sd["score.weight"] = _uni(rng, (2 * H, 2 * H), 1.0 / np.sqrt(2.0 * H))
    _bn(rng, sd, "fc.0", 4 * H)
    # fc_gain > 1 makes the posteriors of a random-weight model less flat so that
    # argmax / beam decisions have a margin (SURVEY.md §7 "hard parts").
    sd["fc.1.weight"] = _uni(rng, (geom.num_class, 4 * H), fc_gain / np.sqrt(4.0 * H))
    return sd


def synth_batch(geom, B, T, L, seed=1234, ragged=True):
    """Stacked features [B,T,feat] f32, canonical ids [B,L] i64 (0-padded),
    float32 length fractions as create_input makes them (data_loader.py:177)."""
    rng = np.random.Generator(np.random.PCG64(seed + 17))
    x = rng.standard_normal((B, T, geom.feat)).astype(np.float32)
    x1 = np.zeros((B, L), dtype=np.int64)
    hi = min(geom.emb_rows, geom.num_class - 1)
    frac = np.ones(B, dtype=np.float32)
    tlen = np.full(B, L, dtype=np.int64)
    for b in range(B):
        lb = L if (b == 0 or not ragged) else int(rng.integers(max(1, L // 2), L + 1))
        x1[b, :lb] = rng.integers(2, hi, size=lb)
        tlen[b] = lb
        if ragged and b > 0:
            tb = int(rng.integers(T // 2, T + 1))
            tb -= tb % 2
            x[b, tb:, :] = 0.0  # collate zero-pads (data_loader.py:159,173)
            frac[b] = np.float32(tb) / np.float32(T)
    return x, x1, frac, tlen


def train_case(geom, seed, B, T, L, Lt, p=0.2):
    """Inputs of one training step (G11 goldens and their tests): seeded weights, batch, one dropout mask per site in the
    reference's tensor layout (1 = keep), padded CTC targets and feasible input lengths ((frac * T').long(), train_ctc.py:68)."""
    sd = synth_state_dict(geom, seed=seed)
    x, x1, frac, _ = synth_batch(geom, B=B, T=T, L=L, seed=seed)
    rs = np.random.Generator(np.random.PCG64(seed + 5))
    Tp = T // 2
    shapes = [(B, geom.channels, T, geom.w1), (B, geom.channels, Tp, geom.w2)] + [(Tp, B, 2 * geom.hidden)] * geom.layers
    masks = [(rs.random(s) >= p).astype(np.uint8) for s in shapes]
    tl = rs.integers(1, Lt + 1, size=B)
    tl[0] = Lt
    tg = np.zeros((B, Lt), dtype=np.int64)
    for b in range(B):
        tg[b, :tl[b]] = rs.integers(1, geom.num_class, size=tl[b])
    il = (frac.astype(np.float32) * np.float32(Tp)).astype(np.int64)
    il = np.minimum(np.maximum(il, 2 * tl + 1), Tp)                      # keep every row feasible
    return sd, x, x1, masks, tg, il.astype(np.int64), tl.astype(np.int64)


def synth_raw_features(B, T_raw=1000, D=81, seed=1234):
    """Raw (CMVN-scale) log-mel + energy frames, N(0,1) f32."""
    rng = np.random.Generator(np.random.PCG64(seed + 29))
    return rng.standard_normal((B, T_raw, D)).astype(np.float32)


def peaky_logp(T, C, n_peaks, seed, blank_boost=6.0, peak_boost=9.0):
    """The 'peaky' posterior generator of SURVEY.md §8(d): logits N(0,1); +9 on one
    random non-blank class at n_peaks random frames, +6 on blank elsewhere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.standard_normal((T, C))
    peaks = rng.choice(T, size=min(n_peaks, T), replace=False)
    is_peak = np.zeros(T, dtype=bool)
    is_peak[peaks] = True
    for t in range(T):
        if is_peak[t]:
            z[t, int(rng.integers(1, C))] += peak_boost
        else:
            z[t, 0] += blank_boost
    z = z.astype(np.float32)
    m = z.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(axis=1, keepdims=True, dtype=np.float32))
    return (z - lse).astype(np.float32)


def phone_table_41():
    """index -> unit for the 41-phone set (45 classes): blank, UNK, sil, 41 phones, err.
    Order follows the reference's table (AA/utils/tools.py:58-104)."""
    units = ("blank UNK sil sh iy hh ae d y er0 aa r k s uw t ih n g w ao dh l ow m eh "
             "oy ay b er v f z th ah ah0 p ey ng ch uh zh jh aw err").split()
    assert len(units) == 45
    return dict(enumerate(units))
