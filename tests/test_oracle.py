"""The CPU oracle (oracle/mdd_oracle.c) against golden vectors made by the reference's own
Python (oracle/gen_golden.py).  CPU only.  This is what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import npz, jload, ids_to_beam_string, ids_to_greedy_string, GOLD
from ctc_attention_mispronunciation_amd import synth

ERR = {0: None, 1: "IndexError", 2: "ValueError", 3: "KeyError"}
# |oracle - reference| budget on log-probs: the reference's own fp32 rounding, far inside the
# 1e-4 parity tolerance BASELINE.json states for the GPU path.
TOL_ORACLE = 2e-5


def test_g1_tiny_every_stage():
    g = npz("g1_tiny.npz")
    geom = synth.Geometry(**synth.TINY)
    sd = synth.synth_state_dict(geom, seed=11)
    taps = {}
    logp = oracle.forward(sd, g["x"], g["x1"], taps)
    for name in ("conv0", "conv1", "rnn0", "rnn1", "text", "key"):
        np.testing.assert_allclose(taps[name], g[name], rtol=0, atol=TOL_ORACLE, err_msg=name)
    np.testing.assert_allclose(taps["logits"].reshape(g["logits"].shape), g["logits"], rtol=0, atol=TOL_ORACLE)
    np.testing.assert_allclose(logp, g["logp"], rtol=0, atol=TOL_ORACLE)
    assert (logp.argmax(-1) == g["logp"].argmax(-1)).all()


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_g2_reference_geometry(idx):
    meta = jload("g2_ref.json")[idx]
    g = npz("g2_ref.npz")
    geom = synth.Geometry(**meta["geom"])
    sd = synth.synth_state_dict(geom, seed=meta["seed"])
    x, x1, frac, tlen = synth.synth_batch(geom, B=meta["B"], T=meta["T"], L=meta["L"], seed=meta["seed"])
    logp = oracle.forward(sd, x, x1)
    ref = g[meta["tag"] + "_logp"]
    assert logp.shape == ref.shape
    np.testing.assert_allclose(logp, ref, rtol=0, atol=TOL_ORACLE)
    if meta["min_top2_gap"] > 1e-3:
        assert (logp.argmax(-1) == ref.argmax(-1)).all()


@pytest.mark.parametrize("si", [0, 1])
def test_g3_decoders(si):
    meta = jload("g3_decode.json")["sets"][si]
    g = npz("g3_decode.npz")
    Cn, i2c = meta["C"], meta["int2char"]
    for r in meta["records"]:
        lp = g["c%d_case%d" % (Cn, r["case"])][:, None, :]
        gr = oracle.greedy(lp, [r["len"]])[0]
        assert ids_to_greedy_string(gr, i2c) == r["greedy"], r
        ids, st = oracle.beam(lp, [r["len"]], g[r["lm"]], beam_width=r["width"], alpha=r["alpha"])
        assert ERR[int(st[0])] == r["error"], r
        if r["error"] is None:
            assert ids_to_beam_string(ids[0], i2c) == r["beam"], r
    for f in meta["failures"]:
        lp = g["c%d_fail_%s" % (Cn, f["name"])][:, None, :]
        ids, st = oracle.beam(lp, [f["len"]], g[f["lm"]], beam_width=f["width"], alpha=f["alpha"])
        assert ERR[int(st[0])] == f["error"], f
        assert ids_to_greedy_string(oracle.greedy(lp, [f["len"]])[0], i2c) == f["greedy"]
    b = meta["batch"]
    batch = g["c%d_batch" % Cn]
    assert [ids_to_greedy_string(x, i2c) for x in oracle.greedy(batch, b["lens"])] == b["greedy"]
    ids, st = oracle.beam(batch, b["lens"], g["lm%d" % Cn])
    assert not st.any()
    assert [ids_to_beam_string(x, i2c) for x in ids] == b["beam"]


def test_g4_alignment():
    g = jload("g4_align.json")
    for r in g["pairs"]:
        vocab = {w: i for i, w in enumerate(sorted(set(r["hyp"] + r["can"])))}
        dist, ops = oracle.align([vocab[w] for w in r["hyp"]], [vocab[w] for w in r["can"]])
        assert dist == r["dist"] and ops == r["ops"], r
    for e in g["empties"]:
        assert e["error"] == "TypeError"
        with pytest.raises(TypeError):
            oracle.align([1] * len(e["s1"].split()), [2] * len(e["s2"].split()))


def test_g8_evaluation_counts():
    """the oracle's batch evaluation against what the reference's own loops counted."""
    g = jload("g8_eval.json")
    keys = ("total", "TA", "FR", "FA", "TRc", "TRw", "total_wer", "num_word")
    for b in g["batches"]:
        assert oracle.eval_counts(b["decoded"], b["labels"], b["canonicals"]) == [b[k] for k in keys], b
        assert oracle.eval_counts(b["decoded_nosil"], b["labels_nosil"], b["canonicals_nosil"]) == [b[k] for k in keys]
    assert g["empty_decode_error"] == "TypeError"
    with pytest.raises(TypeError):
        oracle.eval_counts(["sil"], ["aa b"], ["aa b"])


def test_g5_ctc_loss_and_grad():
    g = npz("g5_ctc.npz")
    for m in jload("g5_ctc.json"):
        i = m["i"]
        nll, grad = oracle.ctc_loss(g["logp%d" % i], g["tg%d" % i], g["il%d" % i], g["tl%d" % i])
        ref_nll, ref_grad = g["nll%d" % i], g["grad%d" % i]
        fin = np.isfinite(ref_nll)
        assert (np.isinf(nll) == ~fin).all()
        np.testing.assert_allclose(nll[fin], ref_nll[fin], rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(grad[:, fin, :], ref_grad[:, fin, :], rtol=0, atol=1e-4)  # ATen keeps the lattice in fp32
        if fin.all():
            assert abs(float(nll.sum()) - m["loss"]) <= 1e-5 * max(1.0, abs(m["loss"]))


def test_g6_stack_skip_and_lengths():
    g = npz("g6_input.npz")
    i = 0
    while "raw%d" % i in g:
        np.testing.assert_array_equal(oracle.stack_skip(g["raw%d" % i]), g["stk%d" % i])
        i += 1
    assert i == 6
    for ln, maxlen, tout, want in g["len_table"]:
        assert oracle.len_frames(ln, maxlen, tout) == want, (ln, maxlen, tout)


# ---------------------------------------------------------------------------- the torch/Python port (CPU baseline)
def test_ref_port_forward_matches_goldens():
    from oracle import ref_port
    g = npz("g2_ref.npz")
    for meta in jload("g2_ref.json"):
        geom = synth.Geometry(**meta["geom"])
        sd = synth.synth_state_dict(geom, seed=meta["seed"])
        x, x1, _, _ = synth.synth_batch(geom, B=meta["B"], T=meta["T"], L=meta["L"], seed=meta["seed"])
        np.testing.assert_allclose(ref_port.forward(sd, x, x1).numpy(), g[meta["tag"] + "_logp"], rtol=0, atol=2e-6)


def test_ref_port_decoders_match_goldens():
    import os
    from oracle import ref_port
    from ctc_attention_mispronunciation_amd.utils.NgramLM import LanguageModel
    from tests.helpers import GOLD
    meta = jload("g3_decode.json")["sets"][1]          # the 9-class set keeps the pure-Python beam quick
    g = npz("g3_decode.npz")
    Cn, i2c = meta["C"], dict(enumerate(meta["int2char"]))
    lm = LanguageModel(os.path.join(GOLD, "lm_synth%d.arpa" % Cn))
    for r in meta["records"]:
        lp = g["c%d_case%d" % (Cn, r["case"])][:, None, :]
        assert ref_port.greedy(lp, [r["len"]], i2c)[0] == r["greedy"]
        assert ref_port.beam(lp, [r["len"]], i2c, lm, r["width"], r["alpha"])[0] == r["beam"], r
    for f in meta["failures"]:
        lmf = LanguageModel(os.path.join(GOLD, f["lm"].replace("lm", "lm_synth") + ".arpa"))
        with pytest.raises({"IndexError": IndexError, "ValueError": ValueError, "KeyError": KeyError}[f["error"]]):
            ref_port.beam(g["c%d_fail_%s" % (Cn, f["name"])][:, None, :], [f["len"]], i2c, lmf, f["width"], f["alpha"])


def test_fbank_oracle_properties_and_fixtures():
    """SURVEY 8(f) #1 (parity with Kaldi unpinned): the numpy restatement behaves like a filterbank -- frame count of
    snip-edges, raw log-energy column, a pure tone lands in the mel bin that contains it -- and reads the reference's
    CMVN statistics file (committed as a data fixture)."""
    import wave
    rs = np.random.Generator(np.random.PCG64(3))
    x = (rs.standard_normal(16000) * 1000).astype(np.float32)
    F = oracle.fbank(x)
    assert F.shape == (1 + (16000 - 400) // 160, 81) and oracle.fbank(x[:399]).shape == (0, 81)
    fr = x[:400] - x[:400].mean(dtype=np.float32)
    np.testing.assert_allclose(F[0, 0], np.log(np.dot(fr, fr)), rtol=1e-6)
    banks = oracle.mel_banks()
    assert len(banks) == 80 and banks[0][0] == 1 and banks[-1][0] + len(banks[-1][1]) == 256
    assert all(0 < w.max() <= 1 and w.min() > 0 for _, w in banks)
    t = np.arange(16000) / 16000.0
    for hz in (300.0, 1000.0, 3500.0):
        tone = oracle.fbank((8000 * np.sin(2 * np.pi * hz * t)).astype(np.float32))
        k = int(round(hz / 31.25))
        want = max(range(80), key=lambda b: banks[b][1][k - banks[b][0]] if 0 <= k - banks[b][0] < len(banks[b][1]) else -1)
        assert int(np.argmax(tone[10, 1:])) == want, (hz, int(np.argmax(tone[10, 1:])), want)
    stats = oracle.read_cmvn_stats(os.path.join(GOLD, "global_fbank_cmvn.txt"))
    assert stats.shape == (2, 82) and stats[0, 81] > 1e6 and stats[1, 81] == 0
    w = wave.open(os.path.join(GOLD, "vocabulary_single_1.wav"))
    wav = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    G = oracle.apply_cmvn(oracle.fbank(wav), stats)
    assert G.shape == (282, 81) and np.isfinite(G).all() and abs(float(G.mean())) < 3 and 0.3 < float(G.std()) < 3


def test_fbank_host_side_wire_formats(tmp_path):
    """Kaldi binary float-matrix ark/scp round trip and the CMVN text reader of the host mirror (no GPU involved)."""
    from ctc_attention_mispronunciation_amd.utils import fbank as fb
    feats = {"utt_a": np.arange(12, dtype=np.float32).reshape(3, 4), "b": np.full((2, 81), 0.5, dtype=np.float32)}
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    fb.write_ark_scp(ark, scp, feats)
    back = fb.read_ark(ark)
    assert list(back) == list(feats) and all(np.array_equal(back[k], feats[k]) for k in feats)
    raw = open(ark, "rb").read()
    for line in open(scp):
        key, loc = line.split()
        off = int(loc.rsplit(":", 1)[1])
        assert raw[off:off + 6] == b"\0BFM \x04"[:6] and raw[off - len(key) - 1:off - 1] == key.encode()
    stats = fb.read_cmvn_stats(os.path.join(GOLD, "global_fbank_cmvn.txt"))
    np.testing.assert_array_equal(stats, oracle.read_cmvn_stats(os.path.join(GOLD, "global_fbank_cmvn.txt")))
    sc, of = fb.cmvn_scale_offset(stats)
    x = np.random.Generator(np.random.PCG64(1)).standard_normal((5, 81)).astype(np.float32)
    np.testing.assert_allclose(x * sc + of, oracle.apply_cmvn(x, stats), rtol=1e-5, atol=1e-5)



@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_g9_benchmarked_length_and_chain(idx):
    """G9: reference log-probs at the benchmarked length (T'=250, ragged, H=384 and H=256) and the reference's whole
    infer.py chain on them (model -> Beam(10)/Greedy -> wer -> align_canonical_decoded -> stastics -> score).  The oracle
    must land within its budget on the log-probs, and -- fed its OWN log-probs, not the reference's -- produce the same
    strings, op paths and scores through the host chain of the product (Decoder.wer = mdd_align, infer_core)."""
    from tests.helpers import chain_inputs, check_chain
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import Decoder
    from ctc_attention_mispronunciation_amd.utils.NgramLM import LanguageModel
    from ctc_attention_mispronunciation_amd.infer_core import diagnose
    meta = jload("g9_chain.json")[idx]
    g = npz("g9_chain.npz")
    geom, sd, x, x1, frac, tlen = chain_inputs(meta)
    if x is None:
        x = g["wav_feats"][None]
    ref = g[meta["tag"] + "_logp"]
    logp = oracle.forward(sd, x, x1)
    np.testing.assert_allclose(logp, ref, rtol=0, atol=TOL_ORACLE)
    i2c = synth.phone_table_41()
    lens = [oracle.lib().orc_len_frames(float(f), ref.shape[0]) for f in frac]
    assert lens == [r["len"] for r in meta["records"]]
    table = LanguageModel(os.path.join(GOLD, "lm_synth45.arpa")).dense_table(i2c, 45)
    ids, st = oracle.beam(logp, lens, table, beam_width=10, alpha=0.0)
    assert not st.any()
    dec = Decoder(i2c, space_idx=-1, blank_index=0)
    check_chain(meta["records"], [ids_to_beam_string(s, i2c) for s in ids],
                [ids_to_greedy_string(s, i2c) for s in oracle.greedy(logp, lens)], dec.wer,
                lambda hyp, can: diagnose(hyp, can, dec))


def test_g12_vocabulary_single_words_through_the_oracle():
    """G12 (the reference's chain on all 20 words of egs/vocabulary/single): the oracle's forward on the same features (oracle.fbank +
    CMVN + stack/skip of each WAV) within 3e-6 of the reference model's posteriors, and its beam / greedy decodes + the host
    alignment and diagnosis identical to the reference's records, for every word that has a canonical."""
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import Decoder
    from ctc_attention_mispronunciation_amd.utils.NgramLM import LanguageModel
    from ctc_attention_mispronunciation_amd.infer_core import diagnose
    import wave
    from tests.helpers import check_chain
    meta, g = jload("g12_words.json"), npz("g12_words.npz")
    geom = synth.Geometry(**synth.REFERENCE)
    sd = synth.synth_state_dict(geom, seed=11)
    i2c = synth.phone_table_41()
    c2i = {v: k for k, v in i2c.items()}
    table = LanguageModel(arpa_file=os.path.join(GOLD, "lm_synth45.arpa")).dense_table(i2c, 45)
    stats = oracle.read_cmvn_stats(os.path.join(GOLD, "global_fbank_cmvn.txt"))
    dec = Decoder(i2c, space_idx=-1, blank_index=0)
    n = 0
    for rec in meta[:8]:                                          # (the oracle's forward is ~2 s per word: a subset keeps the CPU suite short)
        if rec["canonical"] is None:
            continue
        w = wave.open(os.path.join(GOLD, "vocabulary_single", "%d.wav" % rec["i"]))
        wav = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32)
        feats = oracle.stack_skip(oracle.apply_cmvn(oracle.fbank(wav), stats))
        x1 = np.array([[c2i[p] for p in rec["canonical"].split()]], dtype=np.int64)
        logp = oracle.forward(sd, feats[None], x1)
        np.testing.assert_allclose(logp, g["logp%d" % rec["i"]], rtol=0, atol=3e-6)
        lens = [logp.shape[0]]
        ids, st = oracle.beam(logp, lens, table, beam_width=10, alpha=0.0)
        assert not st.any()
        check_chain(rec["records"], [ids_to_beam_string(s, i2c) for s in ids], [ids_to_greedy_string(s, i2c) for s in oracle.greedy(logp, lens)],
                    dec.wer, lambda hyp, can: diagnose(hyp, can, dec))
        n += 1
    assert n >= 6


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_g11_train_step_restatement(idx):
    """G11: the reference model in train mode (batch-statistics BatchNorm, given dropout masks), CTCLoss(sum)/B, backward --
    log-probs, loss, every parameter gradient, updated running statistics.  This pins oracle/ref_port.train_step (the torch
    restatement that checks the HIP training step on shapes without a golden)."""
    from oracle import ref_port
    meta = jload("g11_train.json")[idx]
    g = npz("g11_train.npz")
    tag = meta["tag"]
    geom = synth.Geometry(**meta["geom"])
    sd, x, x1, masks, tg, il, tl = synth.train_case(geom, meta["seed"], meta["B"], meta["T"], meta["L"], meta["Lt"])
    logp, loss, grads, run = ref_port.train_step(sd, x, x1, masks, tg, il, tl, 0.2)
    np.testing.assert_allclose(logp, g[tag + "_logp"], rtol=0, atol=2e-6)
    assert abs(loss - meta["loss"]) <= 1e-5 * abs(meta["loss"])
    assert set(grads) == set(meta["tensors"])
    for k, info in meta["tensors"].items():
        tol = 2e-6 * max(1.0, info["absmax"])
        if tag + "_grad_" + k in g.files:
            np.testing.assert_allclose(grads[k], g[tag + "_grad_" + k], rtol=0, atol=tol, err_msg=k)
        else:
            np.testing.assert_allclose(grads[k].ravel()[g[tag + "_gidx_" + k]], g[tag + "_gval_" + k], rtol=0, atol=tol, err_msg=k)
        assert abs(np.sqrt((grads[k].astype(np.float64) ** 2).sum()) - info["norm"]) <= 1e-4 * max(info["norm"], 1e-6), k
    for k, v in run.items():
        np.testing.assert_allclose(v, g["%s_run_%s" % (tag, k)], rtol=0, atol=1e-6, err_msg=k)
