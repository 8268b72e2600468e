"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package.  See mdd_oracle.c for the reference citations of every function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
EPS = 1e-5  # torch BatchNorm default eps (model_ctc.py:27,60 use the default)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_len_frac.restype = C.c_float
        _LIB.orc_len_frac.argtypes = [C.c_int, C.c_int]
        _LIB.orc_len_frames.argtypes = [C.c_float, C.c_int]
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def stack_skip(raw, right=2, skip=2, n_down=2):
    raw = _f(raw)
    T, D = raw.shape
    To = lib().orc_stack_len(T, skip, n_down)
    out = np.empty((To, (right + 1) * D), dtype=np.float32)
    lib().orc_stack_skip(_p(raw), T, D, right, skip, n_down, _p(out))
    return out


def len_frames(length, maxlen, t_out):
    return lib().orc_len_frames(lib().orc_len_frac(int(length), int(maxlen)), int(t_out))


def conv_bn_relu(x, sd, prefix, stride):
    x = _f(x)
    B, Cin, Hin, Win = x.shape
    w = _f(sd[prefix + ".conv.weight"])
    Cout = w.shape[0]
    Hout, Wout = (Hin + 2 - 3) // stride[0] + 1, (Win + 2 - 3) // stride[1] + 1
    y = np.empty((B, Cout, Hout, Wout), dtype=np.float32)
    bn = [_f(sd[prefix + ".batch_norm." + k]) for k in ("weight", "bias", "running_mean", "running_var")]
    bias = _f(sd[prefix + ".conv.bias"])
    lib().orc_conv_bn_relu(_p(x), B, Cin, Hin, Win, _p(w), _p(bias), _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]),
                           C.c_float(EPS), Cout, stride[0], stride[1], _p(y))
    return y


def bn_rows(x, sd, prefix):
    x = _f(x)
    F = x.shape[-1]
    y = np.empty_like(x)
    bn = [_f(sd[prefix + "." + k]) for k in ("weight", "bias", "running_mean", "running_var")]
    lib().orc_bn_rows(_p(x), C.c_size_t(x.size // F), F, _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), C.c_float(EPS), _p(y))
    return y


def linear(x, W, b=None):
    x, W = _f(x), _f(W)
    K, N = W.shape[1], W.shape[0]
    y = np.empty(x.shape[:-1] + (N,), dtype=np.float32)
    bb = _f(b) if b is not None else None
    lib().orc_linear(_p(x), C.c_size_t(x.size // K), K, _p(W), _p(bb), N, _p(y))
    return y


def bilstm(x, sd, prefix, bias=False):
    """x [T,B,I] -> [T,B,2H]; both directions of one nn.LSTM layer."""
    x = _f(x)
    T, B, I = x.shape
    H = sd[prefix + ".weight_hh_l0"].shape[1]
    out = np.empty((T, B, 2 * H), dtype=np.float32)
    for d, sfx in enumerate(("", "_reverse")):
        wih, whh = _f(sd[prefix + ".weight_ih_l0" + sfx]), _f(sd[prefix + ".weight_hh_l0" + sfx])
        bih = _f(sd[prefix + ".bias_ih_l0" + sfx]) if bias else None
        bhh = _f(sd[prefix + ".bias_hh_l0" + sfx]) if bias else None
        lib().orc_lstm_dir(_p(x), T, B, I, _p(wih), _p(whh), _p(bih), _p(bhh), H, d, _p(out), 2 * H, d * H)
    return out


def attention(X, key, val):
    X, key, val = _f(X), _f(key), _f(val)
    B, T, D = X.shape
    L = key.shape[1]
    out = np.empty((B, T, 2 * D), dtype=np.float32)
    lib().orc_attention(_p(X), _p(key), _p(val), B, T, L, D, _p(out))
    return out


def forward(sd, x, x1, taps=None):
    """CTC_Model.forward(x, x1) (model_ctc.py:160-223), eval mode.  x [B,T,F] f32, x1 [B,L] i64.
    Returns logp [T/2, B, C]; optional dict `taps` receives every stage."""
    x = _f(x)
    B, T, F = x.shape
    a = conv_bn_relu(x.reshape(B, 1, T, F), sd, "conv.0", (1, 2))
    if taps is not None:
        taps["conv0"] = a
    a = conv_bn_relu(a, sd, "conv.1", (2, 2))
    if taps is not None:
        taps["conv1"] = a
    _, Cc, Tp, W = a.shape
    seq = np.empty((Tp, B, Cc * W), dtype=np.float32)
    lib().orc_cnn_to_seq(_p(a), B, Cc, Tp, W, _p(seq))
    n = 0
    while "rnns.%d.rnn.weight_ih_l0" % n in sd:
        if n > 0:
            seq = bn_rows(seq, sd, "rnns.%d.batch_norm" % n)
        seq = bilstm(seq, sd, "rnns.%d.rnn" % n)
        if taps is not None:
            taps["rnn%d" % n] = seq
        n += 1
    X = np.ascontiguousarray(seq.transpose(1, 0, 2))                       # [B,T',2H]
    ids = np.ascontiguousarray(x1, dtype=np.int64)
    L = ids.shape[1]
    emb_t = _f(sd["embeds.weight"])
    emb = np.empty((B, L, emb_t.shape[1]), dtype=np.float32)
    if lib().orc_embed(_p(emb_t), emb_t.shape[0], emb_t.shape[1], _p(ids), C.c_size_t(B * L), _p(emb)) != 0:
        raise IndexError("index out of range in self")
    txt = bilstm(np.ascontiguousarray(emb.transpose(1, 0, 2)), sd, "lstm_embeds", bias=True)   # [L,B,2H]
    val = np.ascontiguousarray(txt.transpose(1, 0, 2))                     # [B,L,2H]
    key = linear(val, sd["score.weight"])
    if taps is not None:
        taps["text"], taps["key"] = val, key
    cat = attention(X, key, val)                                           # [B,T',4H]
    rows = np.ascontiguousarray(cat.transpose(1, 0, 2)).reshape(Tp * B, -1)
    logits = linear(bn_rows(rows, sd, "fc.0"), sd["fc.1.weight"])
    if taps is not None:
        taps["logits"] = logits.copy()
    lib().orc_log_softmax(_p(logits), C.c_size_t(logits.shape[0]), logits.shape[1])
    return logits.reshape(Tp, B, -1)


def greedy(logp, lens, blank=0):
    logp = _f(logp)
    T, B, Cc = logp.shape
    ln = np.ascontiguousarray(lens, dtype=np.int32)
    ids = np.zeros((B, T), dtype=np.int32)
    n = np.zeros(B, dtype=np.int32)
    lib().orc_greedy(_p(logp), T, B, Cc, _p(ln), blank, _p(ids), _p(n))
    return [ids[b, :n[b]].tolist() for b in range(B)]


def beam(logp, lens, lm_table, beam_width=10, alpha=0.0, blank=0, return_scores=False):
    """Returns (list of id lists, status array).  status: 0 ok, 1 IndexError, 2 ValueError, 3 KeyError."""
    logp = _f(logp)
    T, B, Cc = logp.shape
    ln = np.ascontiguousarray(lens, dtype=np.int32)
    lm = np.ascontiguousarray(lm_table, dtype=np.float64)
    assert lm.shape == (Cc + 1, Cc + 1)
    ids = np.zeros((B, T), dtype=np.int32)
    n = np.zeros(B, dtype=np.int32)
    st = np.zeros(B, dtype=np.int32)
    sc = np.zeros(B, dtype=np.float64)
    lib().orc_beam(_p(logp), T, B, Cc, _p(ln), beam_width, blank, _p(lm), C.c_double(alpha), _p(ids), _p(n), _p(st), _p(sc))
    out = [ids[b, :n[b]].tolist() for b in range(B)]
    return (out, st, sc) if return_scores else (out, st)


OPS = "-SID"


def align(a, b):
    """Decoder.wer core on integer tokens: returns (dist, ops list of '-','S','I','D').
    Raises TypeError when either side is empty, as the reference does (ctcDecoder.py:137-138)."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    ops = np.zeros(len(a) + len(b) + 1, dtype=np.uint8)
    dist = C.c_int32(0)
    nops = C.c_int32(0)
    if lib().orc_align(_p(a), len(a), _p(b), len(b), C.byref(dist), _p(ops), C.byref(nops)) != 0:
        raise TypeError("cannot unpack non-iterable int object")
    return dist.value, [OPS[o] for o in ops[:nops.value]]


def ctc_loss(logp, targets, in_len, tgt_len, blank=0, want_grad=True):
    logp = _f(logp)
    T, B, Cc = logp.shape
    tg = np.ascontiguousarray(targets, dtype=np.int64)
    il = np.ascontiguousarray(in_len, dtype=np.int64)
    tl = np.ascontiguousarray(tgt_len, dtype=np.int64)
    nll = np.zeros(B, dtype=np.float32)
    grad = np.zeros_like(logp) if want_grad else None
    lib().orc_ctc_loss(_p(logp), T, B, Cc, _p(tg), tg.shape[1], _p(il), _p(tl), blank, _p(nll), _p(grad))
    return nll, grad


def eval_counts(decoded, labels, canonicals):
    """SURVEY 8(f) #2 -- CPU restatement of the evaluation bookkeeping of AA/steps/test_ctc_nosil.py:33-60,196-209,
    218-298 for one batch of space-separated phoneme strings: returns [canonical phonemes, TA, FR, FA, TR correct,
    TR wrong, summed edit distance decoded-vs-annotated, annotated phonemes].  Small inputs only (pure Python on top of
    `align`).  An empty sequence raises TypeError, as the reference's `_, path = decoder.wer(...)` does."""
    def nosil(s):
        return [t for t in s.split(" ") if t != "sil"]

    def position_map(hyp, can):
        if not hyp or not can:
            raise TypeError("cannot unpack non-iterable int object")
        table = {}
        dist, ops = align([table.setdefault(t, len(table)) for t in hyp], [table.setdefault(t, len(table)) for t in can])
        val, gaps, hi, j = {}, [], 0, 0
        for op in ops:
            if op == "-":
                val[j] = "-"; hi += 1; j += 1
            elif op == "S":
                val[j] = "S" + hyp[hi]; hi += 1; j += 1
            elif op == "D":
                val[j] = "D"; j += 1
            else:
                gaps.append(str(j - 1) + str(j)); hi += 1
        return dist, val, gaps

    out = [0] * 8
    for dec, lab, can in zip(decoded, labels, canonicals):
        dec, lab, can = nosil(dec), nosil(lab), nosil(can)
        _, v1, g1 = position_map(lab, can)
        err, _, _ = position_map(dec, lab)
        _, v2, g2 = position_map(dec, can)
        out[0] += len(can)
        for j in range(len(can)):
            a, b = v1[j], v2[j]
            if a == "-" and b == "-":
                out[1] += 1
            elif a == "-":
                out[2] += 1
            elif b == "-":
                out[3] += 1
            elif a == b:
                out[4] += 1
            else:
                out[5] += 1
        if g1 and not g2:
            out[3] += len(g1)
        elif g2 and not g1:
            out[2] += len(g2)
        elif g1 and g2:
            for e in g1:                      # the reference removes from the list it is iterating: kept, it skips elements
                if e in g2:
                    g1.remove(e)
                    g2.remove(e)
                    out[4] += 1
            out[3] += len(g1)
            out[2] += len(g2)
        out[6] += err
        out[7] += len(lab)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY 8(f) #1 -- the feature front-end the reference shells out to (AA/infer.py:567-574):
#   compute-fbank-feats --config=conf/fbank.conf | apply-cmvn --norm-vars=true data/global_fbank_cmvn.txt
# Kaldi is third-party and not in /root/reference (only unrunnable prebuilt binaries are): this restates Kaldi's
# published algorithm (feat/feature-window.cc ExtractWindow/ProcessWindow, feat/mel-computations.cc MelBanks,
# feat/feature-fbank.cc FbankComputer::Compute, transform/cmvn.cc ApplyCmvn) for the option set of AA/conf/fbank.conf
# (hamming window, 80 mel bins, use-energy) with Kaldi's defaults elsewhere (16 kHz, 25 ms / 10 ms, preemphasis 0.97,
# remove-dc-offset, snip-edges, raw-energy, power spectrum, log).  PARITY UNPINNED: no Kaldi binary or fixture output
# exists here to check it against; dither (Kaldi default 1.0, random) is 0 unless a noise array is passed.
FBANK = dict(samp_freq=16000.0, frame_len=400, frame_shift=160, nfft=512, num_bins=80, low_freq=20.0, preemph=0.97)


def mel_banks(num_bins=80, nfft=512, samp_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """[(first_fft_bin, weights float32[...])] per mel bin (MelBanks::MelBanks, no VTLN, htk_mode=false)."""
    f32 = np.float32
    nyquist = 0.5 * samp_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + f32(f) / f32(700.0), dtype=f32)  # noqa: E731
    bin_width = f32(samp_freq / nfft)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = f32((mel_high - mel_low) / f32(num_bins + 1))
    mel_bins = np.array([mel(bin_width * f32(i)) for i in range(nfft // 2)], dtype=f32)
    out = []
    for b in range(num_bins):
        left = f32(mel_low + f32(b) * delta)
        center = f32(mel_low + f32(b + 1) * delta)
        right = f32(mel_low + f32(b + 2) * delta)
        idx = np.nonzero((mel_bins > left) & (mel_bins < right))[0]
        w = np.where(mel_bins[idx] <= center, (mel_bins[idx] - left) / (center - left), (right - mel_bins[idx]) / (right - center)).astype(f32)
        out.append((int(idx[0]), w))
    return out


def fbank(wav, dither_noise=None):
    """wav: 1-D samples on the int16 scale (Kaldi does not normalise) -> float32 [num_frames, 81]: column 0 = log energy
    of the raw (DC-removed) frame, columns 1..80 = log mel energies."""
    f32 = np.float32
    x = np.asarray(wav, dtype=f32)
    N, S, P = FBANK["frame_len"], FBANK["frame_shift"], FBANK["nfft"]
    nfr = 0 if len(x) < N else 1 + (len(x) - N) // S
    banks = mel_banks()
    window = (f32(0.54) - f32(0.46) * np.cos(2.0 * np.pi * np.arange(N) / (N - 1))).astype(f32)
    eps = np.finfo(np.float32).eps
    out = np.zeros((nfr, 1 + FBANK["num_bins"]), dtype=f32)
    for f in range(nfr):
        fr = x[f * S:f * S + N].astype(f32).copy()
        if dither_noise is not None:
            fr = fr + np.asarray(dither_noise[f], dtype=f32)
        fr = fr - f32(fr.sum(dtype=f32) / f32(N))                       # remove_dc_offset
        out[f, 0] = np.log(max(f32(np.dot(fr, fr)), eps))               # raw_energy: before preemphasis and windowing
        pre = fr.copy()
        pre[1:] = fr[1:] - f32(FBANK["preemph"]) * fr[:-1]
        pre[0] = fr[0] - f32(FBANK["preemph"]) * fr[0]
        pad = np.zeros(P, dtype=f32)
        pad[:N] = pre * window
        spec = np.fft.rfft(pad.astype(np.float64))
        power = (spec.real ** 2 + spec.imag ** 2).astype(f32)           # bins 0..256; MelBanks uses 0..255
        for b, (first, w) in enumerate(banks):
            out[f, 1 + b] = np.log(max(f32(np.dot(w, power[first:first + len(w)])), eps))
    return out


def read_cmvn_stats(path):
    """Kaldi text matrix ' [ a b ...\n c d ... ]' -> float64 [2, D+1] (row 0: sums and count, row 1: sums of squares)."""
    txt = open(path).read().replace("[", " ").replace("]", " ")
    rows = [r.split() for r in txt.strip().split("\n") if r.split()]
    return np.array([[float(v) for v in r] for r in rows], dtype=np.float64)


def apply_cmvn(feats, stats, norm_vars=True):
    """ApplyCmvn with global stats (double arithmetic, variance floor 1e-20), float32 out."""
    D = feats.shape[1]
    count = stats[0, D]
    mean = stats[0, :D] / count
    if norm_vars:
        var = np.maximum(stats[1, :D] / count - mean * mean, 1e-20)
        scale = 1.0 / np.sqrt(var)
    else:
        scale = np.ones(D)
    offset = -mean * scale
    return (feats.astype(np.float64) * scale + offset).astype(np.float32)

