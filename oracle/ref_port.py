"""CPU port of the reference path on the reference's own substrate (torch CPU ops + pure-Python
prefix beam) -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Why a second restatement next to mdd_oracle.c: the C oracle is the arithmetic checker (independent of
ATen, fp64 accumulation), but as a *speed* baseline it would misrepresent the reference both ways (its
scalar network is ~60x slower than ATen's, its C beam ~100x faster than the reference's Python beam).
This module performs the same operations with the same libraries the reference uses -- ATen conv /
LSTM / bmm on the host cores and a dict-based Python beam -- so bench.py's `cpu_baseline` (kind "port")
times what AA/infer.py would cost on the GPU box's host.  Pinned by the same goldens (tests/test_oracle.py).

Reference lines followed: AA/models/model_ctc.py:160-223 (forward), AA/utils/ctcDecoder.py:186-226,
AA/utils/BeamSearch.py:43-153.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_ZERO = -99999999.0


def _bn(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=eps)


def _bilstm(x, sd, prefix, bias, batch_first=False):
    H = sd[prefix + ".weight_hh_l0"].shape[1]
    names = ["weight_ih_l0", "weight_hh_l0"] + (["bias_ih_l0", "bias_hh_l0"] if bias else [])
    flat = [sd["%s.%s%s" % (prefix, n, sfx)] for sfx in ("", "_reverse") for n in names]
    h0 = x.new_zeros(2, x.shape[0] if batch_first else x.shape[1], H)
    out, _, _ = torch._VF.lstm(x, (h0, h0.clone()), flat, bias, 1, 0.0, False, True, batch_first)
    return out


def forward(sd_np, x, x1, dtype=torch.float32, taps=None):
    """Eval-mode CTC_Model.forward on CPU tensors.  sd_np: dict key -> numpy; x [B,T,F]; x1 [B,L] long.  dtype=torch.float64 runs the
    same graph in double (the yardstick for "whose fp32 evaluation is closer": tests/test_gpu_parity.py)."""
    sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v)) for k, v in sd_np.items()}
    x = torch.as_tensor(x, dtype=dtype)
    x1 = torch.as_tensor(x1, dtype=torch.int64)
    with torch.no_grad():
        a = x.unsqueeze(1)
        for n, stride in ((0, (1, 2)), (1, (2, 2))):
            a = F.conv2d(a, sd["conv.%d.conv.weight" % n], sd["conv.%d.conv.bias" % n], stride=stride, padding=(1, 1))
            a = F.relu(_bn(a, sd, "conv.%d.batch_norm" % n))
        B, C, T, W = a.shape
        seq = a.transpose(1, 2).contiguous().view(B, T, C * W).transpose(0, 1).contiguous()
        if taps is not None:
            taps["conv1"] = seq.numpy().copy()
        n = 0
        while "rnns.%d.rnn.weight_ih_l0" % n in sd:
            if n > 0:
                seq = _bn(seq.transpose(-1, -2), sd, "rnns.%d.batch_norm" % n).transpose(-1, -2)
            seq = _bilstm(seq, sd, "rnns.%d.rnn" % n, bias=False)
            if taps is not None:
                taps["rnn%d" % n] = seq.numpy().copy()
            n += 1
        X = seq.transpose(0, 1)
        val = _bilstm(F.embedding(x1, sd["embeds.weight"]), sd, "lstm_embeds", bias=True, batch_first=True)
        key = F.linear(val, sd["score.weight"])
        if taps is not None:
            taps["text"] = val.transpose(0, 1).numpy().copy()      # [L, B, 2H], the library's tap layout
            taps["key"] = key.transpose(0, 1).numpy().copy()
        attn = torch.softmax(torch.bmm(X, key.transpose(1, 2)), dim=-1)
        cat = torch.cat((X, torch.bmm(attn, val)), -1).transpose(0, 1).contiguous()
        Tp = cat.shape[0]
        logits = F.linear(_bn(cat.view(Tp * B, -1), sd, "fc.0"), sd["fc.1.weight"])
        return torch.log_softmax(logits.view(Tp, B, -1), dim=-1)


def greedy(logp, lens, int2char, blank=0):
    best = torch.as_tensor(logp).transpose(0, 1).argmax(2).numpy()
    out = []
    for b, n in enumerate(lens):
        s, prev = "", None
        for t in range(n):
            k = int(best[b, t])
            if k != blank and not (t != 0 and k == prev):
                s += " " + int2char[k]
            prev = k
        out.append(s)
    return out


def _ladd(a, b):
    if a <= LOG_ZERO:
        return b
    if b <= LOG_ZERO:
        return a
    if b - a > 0.0:
        a, b = b, a
    return a + math.log(1 + math.exp(b - a))


def beam(logp, lens, int2char, lm, beam_width=10, alpha=0.0, blank=0):
    """Prefix beam search in pure Python (dict of prefix tuple -> [total, nonblank, blank]) with the
    reference's rules: exp() in fp32 first, skip near-certain blank frames, repeat rule on the raw previous
    frame, stable top-`beam`, EOS LM term, length normalisation.  lm: object with get_bi_prob(w1, w2)."""
    probs = torch.exp(torch.as_tensor(logp).transpose(0, 1)).numpy()
    C = probs.shape[2]
    out = []
    for b in range(probs.shape[0]):
        mat = probs[b]
        last = {(): [0.0, LOG_ZERO, 0.0]}
        for t in range(lens[b]):
            if (1 - mat[t, blank]) < 0.1:
                continue
            keep = sorted(last.items(), key=lambda kv: kv[1][0], reverse=True)[:beam_width]
            curr = {}
            for y, (tot, nb, bl) in keep:
                p_nb = nb + math.log(mat[t, y[-1]]) if y else LOG_ZERO
                p_bl = tot + math.log(mat[t, blank])
                e = curr.setdefault(y, [LOG_ZERO, LOG_ZERO, LOG_ZERO])
                e[1] = _ladd(e[1], p_nb)
                e[2] = _ladd(e[2], p_bl)
                e[0] = _ladd(e[0], _ladd(p_bl, p_nb))
                for k in range(C):
                    if k == blank:
                        continue
                    big = lm.get_bi_prob(int2char[y[-1]] if y else "", int2char[k]) * alpha
                    base = bl if (y and y[-1] == k and mat[t - 1, blank] < 0.9) else tot
                    pr = math.log(mat[t, k]) + big + base
                    e2 = curr.setdefault(y + (k,), [LOG_ZERO, LOG_ZERO, LOG_ZERO])
                    e2[1] = _ladd(e2[1], pr)
                    e2[0] = _ladd(e2[0], pr)
            last = curr
        keep = sorted(last.items(), key=lambda kv: kv[1][0], reverse=True)[:beam_width]
        final = []
        for y, (tot, nb, bl) in keep:
            pr = _ladd(LOG_ZERO, tot + lm.get_bi_prob(int2char[y[-1]], "") * alpha)   # IndexError on the empty prefix
            final.append((pr * (1.0 / (len(y) if len(y) else 1)), y))
        best = sorted(final, key=lambda v: v[0], reverse=True)[0][1]
        out.append(" ".join(int2char[k] for k in best))
    return out


# --------------------------------------------------------------------------------------------------------------------
# Train-mode restatement (torch autograd on ATen CPU ops) -- the checker of the HIP training step on shapes that have no
# golden: CTC_Model.forward with BatchNorm on batch statistics and Dropout realised by GIVEN masks (1 = keep; the kept values are
# scaled by 1/(1-p), as nn.Dropout does), nn.CTCLoss(sum)/B, backward.  Pinned against the real reference by tests/golden/g11_*.
def train_step(sd_np, x, x1, masks, targets, in_len, tgt_len, p_drop, momentum=0.1, eps=1e-5, dtype=torch.float32):
    """Returns (logp [T',B,C], loss, {key: grad}, {running-stat key: updated value}).  masks: list of arrays in the reference's
    layouts: [B,ch,T,W1], [B,ch,T',W2], then [T',B,2H] per BatchRNN layer.  dtype=torch.float64 runs the same graph in double: the
    yardstick for how far two correct fp32 evaluations may differ on an ill-conditioned gradient (tests)."""
    sd = {k: torch.tensor(v, dtype=dtype) for k, v in sd_np.items() if np.asarray(v).dtype.kind == "f"}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if "running_" not in k}
    run = {k: v.clone() for k, v in sd.items() if "running_" in k}
    scale = 1.0 / (1.0 - p_drop)

    def bn(t, prefix):       # t: [N, C, ...], training=True updates the running statistics in place
        return F.batch_norm(t, run[prefix + ".running_mean"], run[prefix + ".running_var"], params[prefix + ".weight"], params[prefix + ".bias"],
                            training=True, momentum=momentum, eps=eps)

    def drop(t, m):
        return t * (torch.as_tensor(m, dtype=dtype) * scale) if p_drop > 0 else t

    def bilstm(t, prefix, bias, batch_first=False):
        H = params[prefix + ".weight_hh_l0"].shape[1]
        names = ["weight_ih_l0", "weight_hh_l0"] + (["bias_ih_l0", "bias_hh_l0"] if bias else [])
        flat = [params["%s.%s%s" % (prefix, n, sfx)] for sfx in ("", "_reverse") for n in names]
        h0 = t.new_zeros(2, t.shape[0] if batch_first else t.shape[1], H)
        out, _, _ = torch._VF.lstm(t, (h0, h0.clone()), flat, bias, 1, 0.0, True, True, batch_first)
        return out

    a = torch.as_tensor(x, dtype=dtype).unsqueeze(1)
    for n, stride in ((0, (1, 2)), (1, (2, 2))):
        a = F.conv2d(a, params["conv.%d.conv.weight" % n], params["conv.%d.conv.bias" % n], stride=stride, padding=(1, 1))
        a = drop(F.relu(bn(a, "conv.%d.batch_norm" % n)), masks[n])
    B, C_, T, W = a.shape
    seq = a.transpose(1, 2).contiguous().view(B, T, C_ * W).transpose(0, 1).contiguous()
    n = 0
    while "rnns.%d.rnn.weight_ih_l0" % n in params:
        if n > 0:
            seq = bn(seq.transpose(-1, -2), "rnns.%d.batch_norm" % n).transpose(-1, -2)
        seq = drop(bilstm(seq, "rnns.%d.rnn" % n, bias=False), masks[2 + n])
        n += 1
    X = seq.transpose(0, 1)
    val = bilstm(F.embedding(torch.as_tensor(x1, dtype=torch.int64), params["embeds.weight"]), "lstm_embeds", bias=True, batch_first=True)
    key = F.linear(val, params["score.weight"])
    attn = torch.softmax(torch.bmm(X, key.transpose(1, 2)), dim=-1)
    cat = torch.cat((X, torch.bmm(attn, val)), -1).transpose(0, 1).contiguous()
    Tp = cat.shape[0]
    logits = F.linear(bn(cat.view(Tp * B, -1), "fc.0"), params["fc.1.weight"])
    logp = torch.log_softmax(logits.view(Tp, B, -1), dim=-1)
    loss = F.ctc_loss(logp, torch.as_tensor(targets, dtype=torch.int64), torch.as_tensor(in_len, dtype=torch.int64),
                      torch.as_tensor(tgt_len, dtype=torch.int64), blank=0, reduction="sum") / B       # train_ctc.py:72-74
    loss.backward()
    return (logp.detach().numpy(), float(loss.item()), {k: v.grad.numpy() for k, v in params.items()}, {k: v.numpy() for k, v in run.items()})
