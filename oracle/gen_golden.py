#!/usr/bin/env python3
"""Golden-vector generator -- TEST INFRASTRUCTURE, runs in the build container only.

Imports the *reference's own* Python (read-only, from /root/reference) with the one
missing third-party module stubbed, feeds it seeded synthetic weights / inputs, and
writes small fixtures under tests/golden/.  The reference never travels to the GPU
box: only these fixtures (data) and this script do.  Nothing from the reference is
copied into the repository.

What is imported:
  AA/models/model_ctc.py     CTC_Model            -> G1 (tiny, every stage), G2 (reference geometry)
  AA/utils/ctcDecoder.py     GreedyDecoder, BeamDecoder, Decoder.wer
  AA/utils/BeamSearch.py     ctcBeamSearch        -> G3 (decoder vectors incl. failure modes)
  AA/utils/NgramLM.py        LanguageModel        -> G3 LM tables
  AA/utils/tools.py          make_context, skip_feat -> G6
  AA/infer.py                align_canonical_decoded / stastics / print_aligned_string
                             (function definitions only, extracted with ast because the
                             module has import-time argparse + absent deps) -> G7
  all of the above chained as AA/infer.py:294-342 does, at T'=250 -> G9 (g9_chain.*)
  the same chain for each of the 20 words of egs/vocabulary/single, canonicals from AA/dict/phonetic_dict.py's CMU lookup -> G12
  torch.nn.CTCLoss(reduction='sum') as called at AA/steps/train_ctc.py:72,186 -> G5

Usage:  python oracle/gen_golden.py            (writes tests/golden/*)
"""
import ast
import json
import math
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AA = "/root/reference/egs/attention_aug"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "ctc-attention-mispronunciation_amd"))
import synth  # noqa: E402
sys.path.pop(0)   # the package mirrors the reference's module names (models/, utils/): keep it off the path from here on

if not os.path.isdir(AA):
    sys.exit("reference tree not present: goldens can only be regenerated in the build container")
sys.modules.setdefault("editdistance", types.ModuleType("editdistance"))
sys.path.insert(0, AA)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from models.model_ctc import CTC_Model  # noqa: E402  (reference)
from utils.ctcDecoder import GreedyDecoder, BeamDecoder  # noqa: E402  (reference)
from utils.NgramLM import LanguageModel  # noqa: E402  (reference)
import utils.tools as ref_tools  # noqa: E402  (reference)

torch.set_num_threads(8)


def build_reference_model(geom, sd_np):
    m = CTC_Model(add_cnn=True, cnn_param=geom.cnn_param(nn), rnn_param=geom.rnn_param(nn),
                  num_class=geom.num_class, drop_out=0.2)
    if geom.emb_rows != 44 or geom.emb_dim != 512:
        # tiny geometry only: the reference hard-codes Embedding(44,512) (model_ctc.py:149-150);
        # swap in same-typed modules of the small shape so every other line of forward() is the reference's.
        m.embeds = nn.Embedding(geom.emb_rows, geom.emb_dim)
        m.lstm_embeds = nn.LSTM(geom.emb_dim, geom.hidden, batch_first=True, bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    m.eval()
    return m


# ----------------------------------------------------------------------------- G1 / G2
def gen_model_goldens():
    # G1: tiny geometry, every stage
    geom = synth.Geometry(**synth.TINY)
    sd = synth.synth_state_dict(geom, seed=11)
    x, x1, frac, tlen = synth.synth_batch(geom, B=3, T=12, L=4, seed=11)
    m = build_reference_model(geom, sd)
    taps = {}

    def hook(name):
        def f(mod, inp, out):
            taps[name] = (out[0] if isinstance(out, tuple) else out).detach().numpy().copy()
        return f
    m.conv[0].register_forward_hook(hook("conv0"))
    m.conv[1].register_forward_hook(hook("conv1"))
    for i in range(geom.layers):
        m.rnns[i].register_forward_hook(hook("rnn%d" % i))
    m.lstm_embeds.register_forward_hook(hook("text"))
    m.score.register_forward_hook(hook("key"))
    m.fc.register_forward_hook(hook("logits"))
    with torch.no_grad():
        logp = m(torch.from_numpy(x), torch.from_numpy(x1)).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_tiny.npz"), x=x, x1=x1, frac=frac, logp=logp, **taps)
    print("G1 tiny: logp", logp.shape, "taps", sorted(taps))

    # G2: reference geometry, logp only (weights regenerated from the seed on both sides)
    out = {}
    meta = []
    for tag, g, B, T, L, seed in (
            ("h384c45", dict(synth.REFERENCE), 2, 64, 7, 1234),
            ("h384c43", dict(synth.REFERENCE, num_class=43), 3, 48, 9, 77),
            ("h256c45", dict(synth.REFERENCE_256), 2, 64, 5, 4321)):
        geom = synth.Geometry(**g)
        sd = synth.synth_state_dict(geom, seed=seed)
        x, x1, frac, tlen = synth.synth_batch(geom, B=B, T=T, L=L, seed=seed)
        m = build_reference_model(geom, sd)
        with torch.no_grad():
            logp = m(torch.from_numpy(x), torch.from_numpy(x1)).numpy()
        out[tag + "_logp"] = logp
        srt = np.sort(logp, axis=-1)
        meta.append(dict(tag=tag, geom=g, B=B, T=T, L=L, seed=seed,
                         min_top2_gap=float((srt[..., -1] - srt[..., -2]).min()),
                         n_params=int(sum(p.numel() for p in m.parameters()))))
        print("G2", tag, logp.shape, "min top-2 gap %.3e" % meta[-1]["min_top2_gap"])
    np.savez_compressed(os.path.join(OUT, "g2_ref.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "g2_ref.json"), "w"), indent=1)


# ----------------------------------------------------------------------------- G3
def write_synth_arpa(path, units, seed, drop=()):
    """ARPA in the layout NgramLM.initngrams parses (tab-separated; 2- or 3-column rows)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    words = ["<s>", "</s>", "<unk>"] + [u for u in units if u not in drop]
    bigr = []
    for w1 in words:
        if w1 == "</s>":
            continue
        for w2 in words:
            if w2 == "<s>" or rng.random() < 0.45:
                continue
            bigr.append((w1, w2, -float(rng.uniform(0.2, 3.0))))
    with open(path, "w") as f:
        f.write("\n\\data\\\nngram  1=%10d\nngram  2=%10d\n\n\\1-grams:\n" % (len(words), len(bigr)))
        for w in words:
            lp = -float(rng.uniform(0.5, 3.5))
            if w == "</s>":
                f.write("%.6f\t%s\n" % (lp, w))
            else:
                f.write("%.6f\t%s\t%.6f\n" % (lp, w, -float(rng.uniform(0.05, 1.2))))
        f.write("\n\\2-grams:\n")
        for w1, w2, lp in bigr:
            f.write("%.6f\t%s %s\n" % (lp, w1, w2))
        f.write("\n\\end\\\n")


def lm_table(lm, int2char, C):
    """Dense table the C-ABI takes: T[prev][next], prev in 0..C (C = sentence start ''),
    next in 0..C (C = sentence end '').  NaN where the reference would raise KeyError."""
    tab = np.full((C + 1, C + 1), np.nan, dtype=np.float64)
    for p in range(C + 1):
        for n in range(C + 1):
            c1 = "" if p == C else int2char[p]
            c2 = "" if n == C else int2char[n]
            try:
                tab[p, n] = lm.get_bi_prob(c1, c2)
            except KeyError:
                pass
    return tab


def logsoftmax32(z):
    return torch.log_softmax(torch.from_numpy(np.asarray(z, dtype=np.float32)), dim=-1).numpy()


def decode_cases(C):
    """Seeded posterior sets covering the quirks of SURVEY.md §8(a) rows A8/A9."""
    cases = []
    rs = np.random.Generator(np.random.PCG64(99 + C))
    for i in range(14):  # peaky, several densities and lengths
        T = int(rs.integers(20, 81))
        cases.append(("peaky", synth.peaky_logp(T, C, n_peaks=int(rs.integers(2, T // 3 + 2)), seed=1000 + i + C)))
    for i in range(6):   # flat: nothing skipped, beam worst case
        T = int(rs.integers(8, 31))
        cases.append(("flat", logsoftmax32(rs.standard_normal((T, C)))))
    for i in range(6):   # tie-heavy: logits on a 0.5 grid -> many exactly equal probabilities
        T = int(rs.integers(8, 25))
        z = np.round(rs.standard_normal((T, C)) * 2.0) / 2.0
        cases.append(("ties", logsoftmax32(z)))
    for i in range(8):   # repeats straddling the 0.9 / 0.1 thresholds, with skipped-blank runs between
        T = int(rs.integers(16, 41))
        z = rs.standard_normal((T, C)) * 0.3
        t = 0
        k = int(rs.integers(1, C))
        while t < T:
            mode = int(rs.integers(0, 4))
            if mode == 0:      # strong symbol (same as previous half of the time -> repeat rule)
                if rs.random() < 0.5:
                    k = int(rs.integers(1, C))
                z[t, k] += 7.0
            elif mode == 1:    # blank near the 0.9 threshold
                z[t, 0] += float(rs.choice([5.6, 5.9, 6.0, 6.1, 6.4]))
            elif mode == 2:    # near-certain blank (skipped frame)
                z[t, 0] += 12.0
            else:              # blank just around 1-p<0.1
                z[t, 0] += float(rs.choice([5.95, 6.0, 6.05]))
            t += 1
        cases.append(("repeat", logsoftmax32(z)))
    # thresholds hit exactly in float32: p_blank == 0.9 and 1-p_blank == 0.1
    for pb in (0.9, 0.90000004, 0.89999998, 0.1):
        T = 6
        p = np.full((T, C), (1.0 - pb) / (C - 1), dtype=np.float64)
        p[:, 0] = pb
        p[1, :] = 0.2 / (C - 1); p[1, 3] = 0.8 + 0.2 / (C - 1) - 0.2 / (C - 1) * 1  # symbol frame
        p[1] /= p[1].sum()
        p[3] = p[1]
        cases.append(("thresh", np.log(p).astype(np.float32)))
    return cases


def gen_decode_goldens():
    arrays = {}
    meta = {"sets": []}
    for C, units_tab in ((45, synth.phone_table_41()), (9, dict(enumerate("blank UNK sil aa b k iy s t".split())))):
        int2char = units_tab
        units = [int2char[i] for i in range(2, C)]
        arpa = os.path.join(OUT, "lm_synth%d.arpa" % C)
        write_synth_arpa(arpa, ["UNK"] + units, seed=5 + C)
        arpa_missing = os.path.join(OUT, "lm_synth%d_missing.arpa" % C)
        write_synth_arpa(arpa_missing, ["UNK"] + units, seed=6 + C, drop=(int2char[C - 2],))
        lm = LanguageModel(arpa_file=arpa)
        arrays["lm%d" % C] = lm_table(lm, int2char, C)
        arrays["lm%d_missing" % C] = lm_table(LanguageModel(arpa_file=arpa_missing), int2char, C)
        greedy = GreedyDecoder(int2char, space_idx=-1, blank_index=0)
        cases = decode_cases(C)
        recs = []
        for alpha, width in ((0.0, 10), (0.35, 10), (0.0, 3), (0.2, 1)):
            beam = BeamDecoder(int2char, beam_width=width, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=alpha)
            for ci, (kind, lp) in enumerate(cases):
                T = lp.shape[0]
                rs = np.random.Generator(np.random.PCG64(ci * 7 + width))
                ln = T if ci % 3 else int(rs.integers(max(1, T // 2), T + 1))
                pt = torch.from_numpy(lp).unsqueeze(1)  # [T,1,C]
                g = greedy.decode(pt, [ln])[0]
                try:
                    b = beam.decode(pt, [ln])[0]
                    err = None
                except (IndexError, ValueError, KeyError) as e:
                    b, err = None, type(e).__name__
                recs.append(dict(case=ci, kind=kind, len=ln, alpha=alpha, width=width, lm="lm%d" % C,
                                 greedy=g, beam=b, error=err))
        for ci, (kind, lp) in enumerate(cases):
            arrays["c%d_case%d" % (C, ci)] = lp
        # failure modes
        fails = []
        T = 12
        allblank = logsoftmax32(np.concatenate([np.full((T, 1), 14.0), np.zeros((T, C - 1))], axis=1))
        z = np.random.Generator(np.random.PCG64(3)).standard_normal((T, C)); z[:, 2] = -300.0
        underflow = logsoftmax32(z)
        fewlive = allblank.copy(); fewlive[4] = logsoftmax32(np.zeros((1, C)))[0]
        normal = synth.peaky_logp(T, C, 4, seed=8)
        for name, lp, path, alpha in (("allblank", allblank, arpa, 0.0), ("underflow", underflow, arpa, 0.0),
                                      ("fewlive", fewlive, arpa, 0.0), ("missing_unigram", normal, arpa_missing, 0.0),
                                      ("missing_unigram_alpha", normal, arpa_missing, 0.5)):
            beam = BeamDecoder(int2char, beam_width=10, blank_index=0, space_idx=-1, lm_path=path, lm_alpha=alpha)
            try:
                b = beam.decode(torch.from_numpy(lp).unsqueeze(1), [T])[0]
                err = None
            except (IndexError, ValueError, KeyError) as e:
                b, err = None, type(e).__name__
            arrays["c%d_fail_%s" % (C, name)] = lp
            fails.append(dict(name=name, len=T, alpha=alpha, width=10,
                              lm=("lm%d_missing" % C) if path == arpa_missing else "lm%d" % C,
                              greedy=greedy.decode(torch.from_numpy(lp).unsqueeze(1), [T])[0], beam=b, error=err))
        # a real batch through the batch API ([T,B,C] + list of lengths), padded with garbage beyond len
        Tm = max(lp.shape[0] for _, lp in cases[:8])
        batch = np.zeros((Tm, 8, C), dtype=np.float32)
        lens = []
        for b_, (_, lp) in enumerate(cases[:8]):
            batch[:, b_, :] = logsoftmax32(np.random.Generator(np.random.PCG64(b_)).standard_normal((Tm, C)))
            batch[:lp.shape[0], b_, :] = lp
            lens.append(lp.shape[0])
        beam = BeamDecoder(int2char, beam_width=10, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)
        arrays["c%d_batch" % C] = batch
        meta["sets"].append(dict(C=C, int2char=[int2char[i] for i in range(C)], n_cases=len(cases), records=recs,
                                 failures=fails,
                                 batch=dict(lens=lens, greedy=greedy.decode(torch.from_numpy(batch), lens),
                                            beam=beam.decode(torch.from_numpy(batch), lens))))
        nerr = sum(1 for r in recs if r["error"])
        print("G3 C=%d: %d cases x 4 configs, %d raised; failures:" % (C, len(cases), nerr),
              [(f["name"], f["error"]) for f in fails])
    np.savez_compressed(os.path.join(OUT, "g3_decode.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "g3_decode.json"), "w"))


# ----------------------------------------------------------------------------- G4 / G7
def load_infer_functions():
    """exec only the pure helper function definitions of AA/infer.py (no module-level code)."""
    src = open(os.path.join(AA, "infer.py")).read()
    tree = ast.parse(src)
    want = {"align_canonical_decoded", "stastics", "print_aligned_string"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"math": math}
    exec(compile(ast.Module(body=body, type_ignores=[]), "infer.py<helpers>", "exec"), ns)
    return ns


def gen_align_goldens():
    dec = GreedyDecoder(synth.phone_table_41(), space_idx=-1, blank_index=0)
    helpers = load_infer_functions()
    units = [synth.phone_table_41()[i] for i in range(3, 44)]
    rs = np.random.Generator(np.random.PCG64(2024))
    recs = []
    pairs = []
    for i in range(220):
        n2 = int(rs.integers(1, 14))
        can = [units[int(j)] for j in rs.integers(0, 12 if i % 2 else len(units), size=n2)]
        hyp = list(can)
        for _ in range(int(rs.integers(0, 5))):   # mutate: sub / ins / del
            op = int(rs.integers(0, 3))
            if op == 0 and hyp:
                hyp[int(rs.integers(0, len(hyp)))] = units[int(rs.integers(0, len(units)))]
            elif op == 1:
                hyp.insert(int(rs.integers(0, len(hyp) + 1)), units[int(rs.integers(0, len(units)))])
            elif hyp and len(hyp) > 1:
                del hyp[int(rs.integers(0, len(hyp)))]
        if i % 10 == 0:
            hyp = [units[int(j)] for j in rs.integers(0, len(units), size=int(rs.integers(1, 16)))]
        pairs.append((hyp, can))
    pairs += [(["aa"], ["aa"]), (["aa"], ["b"]), (["aa", "aa", "b"], ["aa", "b"]), (["b", "aa", "b"], ["aa"]),
              (["aa"], ["b", "aa", "k", "s"]), (["k", "k", "k"], ["k"]), (["s", "aa", "aa", "t"], ["aa", "t"])]
    for hyp, can in pairs:
        s1, s2 = " ".join(hyp), " ".join(can)
        dist, ops = dec.wer(s1, s2)
        a1, a2, al = helpers["align_canonical_decoded"](list(hyp), list(can), list(ops))
        ins, sub, dele = helpers["stastics"](al, a2, a1)
        ds = sum(1 for c in al if c in "DS")
        cor = sum(1 for c in al if c == "-")
        tmp = min(len(ins) / 4, 0.1 * (cor + ds))
        score = math.ceil((1 - (ds + tmp) / (ds + cor)) * 100) if (ds + cor) else None  # infer.py:338-342
        recs.append(dict(hyp=hyp, can=can, dist=int(dist), ops=ops, al_hyp=a1, al_can=a2, al_ops=al,
                         ins=ins, sub=sub, dele=dele, correct=cor, del_sub=ds, score=score,
                         printed=list(helpers["print_aligned_string"](a1, a2, al))))
    # empty-side behaviour (TypeError, ctcDecoder.py:137-138)
    empties = []
    for s1, s2 in (("", "aa b"), ("aa b", ""), ("", "")):
        try:
            dec.wer(s1, s2)
            empties.append(dict(s1=s1, s2=s2, error=None))
        except Exception as e:  # noqa: BLE001
            empties.append(dict(s1=s1, s2=s2, error=type(e).__name__))
    json.dump(dict(pairs=recs, empties=empties), open(os.path.join(OUT, "g4_align.json"), "w"))
    print("G4/G7 align: %d pairs; empties ->" % len(recs), [e["error"] for e in empties])


# ----------------------------------------------------------------------------- G8 (batch evaluation counts)
def load_eval_pieces():
    """From AA/steps/test_ctc_nosil.py take, by ast and without running the module: the helper
    `print_align_space_canonical_origin` (:33-82) and, out of `test()`, the two per-batch loops that strip 'sil'
    (:196-209) and that count TA / FR / FA / TR (:218-298).  They are executed as they stand, in a namespace we fill."""
    src = open(os.path.join(AA, "steps", "test_ctc_nosil.py")).read()
    tree = ast.parse(src)
    helper = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "print_align_space_canonical_origin"]
    test_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "test"][0]
    loops = {}
    for node in ast.walk(test_fn):
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Call) and isinstance(node.target, ast.Name):
            txt = ast.unparse(node.iter)
            if node.target.id == "i" and txt == "range(len(labels))":
                loops["strip"] = node
            if node.target.id == "x" and txt == "range(len(labels_nosil))":
                loops["count"] = node
    assert set(loops) == {"strip", "count"}, loops.keys()
    ns = {}
    exec(compile(ast.Module(body=helper, type_ignores=[]), "test_ctc_nosil.py<helper>", "exec"), ns)
    code = {k: compile(ast.Module(body=[v], type_ignores=[]), "test_ctc_nosil.py<%s loop>" % k, "exec") for k, v in loops.items()}
    return ns, code


def gen_eval_goldens():
    dec = GreedyDecoder(synth.phone_table_41(), space_idx=-1, blank_index=0)
    ns, code = load_eval_pieces()
    units = [synth.phone_table_41()[i] for i in range(3, 44)]
    rs = np.random.Generator(np.random.PCG64(77))

    def mutate(seq, nmax):
        out = list(seq)
        for _ in range(int(rs.integers(0, nmax + 1))):
            op = int(rs.integers(0, 3))
            if op == 0 and out:
                out[int(rs.integers(0, len(out)))] = units[int(rs.integers(0, len(units)))]
            elif op == 1:
                pos = int(rs.integers(0, len(out) + 1))
                for _r in range(int(rs.integers(1, 3))):           # runs of insertions: repeated gap keys
                    out.insert(pos, units[int(rs.integers(0, len(units)))])
            elif len(out) > 1:
                del out[int(rs.integers(0, len(out)))]
        return out

    def with_sil(seq):
        out = list(seq)
        for _ in range(int(rs.integers(0, 3))):
            out.insert(int(rs.integers(0, len(out) + 1)), "sil")
        return out

    batches = []
    for bi in range(12):
        n = int(rs.integers(3, 9))
        can, lab, hyp = [], [], []
        for _ in range(n):
            c = [units[int(j)] for j in rs.integers(0, 14 if bi % 2 else len(units), size=int(rs.integers(2, 16)))]
            l = mutate(c, 3)                      # what the speaker actually said (annotation)
            h = mutate(l if rs.random() < 0.7 else c, 3)   # what the model heard
            can.append(with_sil(c)); lab.append(with_sil(l)); hyp.append(with_sil(h))
        env = dict(ns)
        env.update(decoder=dec, labels=[" ".join(v) for v in lab], decoded=[" ".join(v) for v in hyp],
                   canonicals=[" ".join(v) for v in can], decoded_nosil=[], labels_nosil=[], canonicals_nosil=[])
        exec(code["strip"], env)
        env.update(utt_list=["UTT%02d_%02d" % (bi, i) for i in range(n)], m_speaker=[], test_wrd_dict={},
                   total_phonemes_in_canonical=0, m_total_phonemes_in_canonical=0, true_accept=0, m_true_accept=0,
                   false_rejection=0, m_false_rejection=0, false_accept=0, m_false_accept=0,
                   true_rejection_correct_diagnose=0, m_true_rejection_correct_diagnose=0,
                   true_rejection_wrong_diagnose=0, m_true_rejection_wrong_diagnose=0, total_wer=0, m_wer=0, m_decoder_num=0)
        dec.num_word = 0
        exec(code["count"], env)
        rec = dict(decoded=env["decoded"], labels=env["labels"], canonicals=env["canonicals"],
                   decoded_nosil=env["decoded_nosil"], labels_nosil=env["labels_nosil"], canonicals_nosil=env["canonicals_nosil"],
                   total=int(env["total_phonemes_in_canonical"]), TA=int(env["true_accept"]), FR=int(env["false_rejection"]),
                   FA=int(env["false_accept"]), TRc=int(env["true_rejection_correct_diagnose"]),
                   TRw=int(env["true_rejection_wrong_diagnose"]), total_wer=int(env["total_wer"]), num_word=int(dec.num_word))
        tr = rec["TRc"] + rec["TRw"]
        try:                                       # test_ctc_nosil.py:310-314
            pr = float(tr) / (tr + rec["FR"]); rc = float(tr) / (tr + rec["FA"])
            rec.update(precision=pr * 100, recall=rc * 100, f1=2 * pr * rc / (pr + rc) * 100, error=None)
        except ZeroDivisionError:
            rec.update(precision=None, recall=None, f1=None, error="ZeroDivisionError")
        batches.append(rec)
    # the helper alone, on a few hand-made paths (keys of the 'I' list, 'S' + phone values)
    singles = []
    for hyp, can in ((["aa", "b"], ["aa", "b"]), (["k", "aa", "b", "s"], ["aa", "b"]), (["aa"], ["aa", "b", "k"]),
                     (["s", "s", "aa", "t", "t"], ["aa", "k", "t"]), (["b"] * 12, ["b"] * 11 + ["k"])):
        s1, s2 = " ".join(hyp), " ".join(can)
        _, path = dec.wer(s1, s2)
        a, b, c, d = ns["print_align_space_canonical_origin"](s1, s2, list(path))
        singles.append(dict(s1=s1, s2=s2, path=path, out=[a, b, c], d={str(k): v for k, v in d.items()}))
    # an utterance whose decode is empty: the reference's loop dies with TypeError (ctcDecoder.py:137-138)
    try:
        dec.wer("", "aa b")
        empty_err = None
    except Exception as e:  # noqa: BLE001
        empty_err = type(e).__name__
    json.dump(dict(batches=batches, singles=singles, empty_decode_error=empty_err), open(os.path.join(OUT, "g8_eval.json"), "w"))
    print("G8 eval: %d batches; totals TA/FR/FA/TRc/TRw =" % len(batches),
          [sum(b[k] for b in batches) for k in ("TA", "FR", "FA", "TRc", "TRw")], "empty ->", empty_err)


# ----------------------------------------------------------------------------- G5
def gen_ctc_goldens():
    arrays = {}
    meta = []
    rs = np.random.Generator(np.random.PCG64(555))
    for i in range(12):
        C = 45 if i % 2 == 0 else 9
        T = int(rs.integers(6, 50))
        B = int(rs.integers(1, 6))
        Lmax = int(rs.integers(1, max(2, min(12, T // 2))))
        logits = rs.standard_normal((T, B, C)).astype(np.float32) * (1.0 + (i % 3))
        tl = rs.integers(1, Lmax + 1, size=B)
        tl[0] = Lmax
        tg = np.zeros((B, Lmax), dtype=np.int64)
        for b in range(B):
            row = rs.integers(1, C, size=tl[b])
            if i % 4 == 1 and tl[b] > 1:       # force repeats (need a blank between)
                row[1] = row[0]
            tg[b, :tl[b]] = row
        il = rs.integers(T // 2 + 1, T + 1, size=B)
        il[0] = T
        for b in range(B):                      # keep most cases feasible: T >= L + repeats
            il[b] = max(il[b], min(T, 2 * tl[b] + 1))
        if i == 7:                              # one infeasible row -> inf loss, as the reference would produce
            il[B - 1] = 1; tl[B - 1] = min(Lmax, 3); tg[B - 1, :tl[B - 1]] = [3, 3, 4][:tl[B - 1]]
        lp = torch.log_softmax(torch.from_numpy(logits), dim=-1).requires_grad_(True)
        loss_fn = nn.CTCLoss(reduction="sum")   # AA/steps/train_ctc.py:186
        loss = loss_fn(lp, torch.from_numpy(tg), torch.from_numpy(il.astype(np.int64)), torch.from_numpy(tl.astype(np.int64)))
        per = nn.CTCLoss(reduction="none")(lp, torch.from_numpy(tg), torch.from_numpy(il.astype(np.int64)),
                                           torch.from_numpy(tl.astype(np.int64))).detach().numpy()
        loss.backward()
        arrays["logp%d" % i] = lp.detach().numpy()
        arrays["tg%d" % i] = tg
        arrays["il%d" % i] = il.astype(np.int64)
        arrays["tl%d" % i] = tl.astype(np.int64)
        arrays["nll%d" % i] = per
        arrays["grad%d" % i] = lp.grad.numpy()
        meta.append(dict(i=i, C=C, T=T, B=B, Lmax=Lmax, loss=float(loss.item())))
    np.savez_compressed(os.path.join(OUT, "g5_ctc.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "g5_ctc.json"), "w"), indent=1)
    print("G5 ctc:", [round(m["loss"], 3) for m in meta])


# ----------------------------------------------------------------------------- G6
def gen_input_goldens():
    arrays = {}
    rs = np.random.Generator(np.random.PCG64(31))
    shapes = [(7, 5), (10, 81), (1, 3), (2, 4), (33, 6), (64, 81)]
    for i, (T, D) in enumerate(shapes):
        raw = rs.standard_normal((T, D))           # kaldiio hands float arrays; reference math is numpy
        st = ref_tools.skip_feat(ref_tools.make_context(raw, 0, 2), 2)      # data_loader.py:138
        if st.shape[0] % 2:                                                  # data_loader.py:140-142
            st = np.vstack([st, np.zeros((2 - st.shape[0] % 2, st.shape[1]))])
        arrays["raw%d" % i] = raw.astype(np.float32)
        arrays["stk%d" % i] = torch.from_numpy(st).float().numpy()           # create_input: .float()
    # float32 length bookkeeping: frac = len/maxlen (f32), then (frac * T_out).long()  (data_loader.py:177, infer.py:296-297)
    rows = []
    for maxlen in (2, 6, 50, 98, 250, 500, 501 * 2, 1234):
        for ln in sorted(set([2, maxlen // 3 * 2 // 2 * 2 or 2, maxlen - 2 if maxlen > 2 else 2, maxlen])):
            sizes = torch.zeros(1)
            sizes[0] = ln / maxlen
            sizes = sizes.float()
            for tout in (maxlen // 2, maxlen // 2 + 1):
                rows.append((ln, maxlen, tout, int((sizes * tout).long()[0])))
    arrays["len_table"] = np.array(rows, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g6_input.npz"), **arrays)
    print("G6 input: %d stack cases, %d length rows" % (len(shapes), len(rows)))


# ----------------------------------------------------------------------------- G9 (long goldens + end-to-end chain)
CHAIN_CASES = (
    # tag, geometry, B, T (stacked frames), L, seed     -- T=500 is the benchmarked length (T'=250 posterior frames)
    ("h384_t500", dict(synth.REFERENCE), 3, 500, 40, 2025),
    ("h256_t500", dict(synth.REFERENCE_256), 2, 500, 40, 2026),
    ("h384_t64", dict(synth.REFERENCE), 4, 64, 7, 2027),
)


def chain_records(logp, frac, x1, tlen, int2char, beam, greedy, helpers):
    """The loop of AA/infer.py:294-342 on one batch of posteriors, run with the reference's own decoders, wer,
    align_canonical_decoded and stastics; returns one record per utterance."""
    probs = torch.from_numpy(logp)
    lens = (torch.from_numpy(frac) * probs.size(0)).long().numpy().tolist()          # infer.py:296-297
    dec_beam = beam.decode(probs, lens)                                              # what infer.py runs (decode_type != 'Greedy')
    dec_greedy = greedy.decode(probs, lens)
    recs = []
    for b in range(len(lens)):
        canonical = " ".join(int2char[int(n)] for n in x1[b][:tlen[b]])              # infer.py:303-306
        rec = dict(len=lens[b], canonical=canonical, beam=dec_beam[b], greedy=dec_greedy[b])
        for name, hyp in (("beam", dec_beam[b]), ("greedy", dec_greedy[b])):
            hyp_ns = " ".join(p for p in hyp.split(" ") if p != "sil")               # :311-318
            can_ns = " ".join(p for p in canonical.split(" ") if p != "sil")
            hyp_ns = hyp_ns.replace("err", "").replace("  ", " ")                    # :322-323
            try:
                dist, path = greedy.wer(hyp_ns, can_ns)                              # :324
            except TypeError:                                                        # an empty side: ctcDecoder.py:137-138
                rec[name + "_chain"] = dict(error="TypeError")
                continue
            pd = [c for c in hyp_ns.split(" ") if c]
            pc = [c for c in can_ns.split(" ") if c]
            pd, pc, path2 = helpers["align_canonical_decoded"](pd, pc, list(path))   # :333
            ds = sum(1 for c in path2 if c in "DS")
            cor = sum(1 for c in path2 if c == "-")
            ins, sub, dele = helpers["stastics"](path2, pc, pd)                      # :338
            tmp = min(len(ins) / 4, 0.1 * (cor + ds))
            score = math.ceil((1 - (ds + tmp) / (ds + cor)) * 100)                   # :339-340
            rec[name + "_chain"] = dict(dist=int(dist), ops=list(path), al_hyp=pd, al_can=pc, al_ops=path2, ins=ins, sub=sub,
                                        dele=dele, correct=cor, del_sub=ds, score=score)
        recs.append(rec)
    return recs


def gen_chain_goldens():
    """G2-long: reference log-probs at the benchmarked length (T'=250), ragged lengths and ragged L, H=384 and H=256;
    G9: the reference's whole chain on them: model -> BeamDecoder(10) / GreedyDecoder -> wer -> align -> stastics -> score.
    Also the WAV of egs/vocabulary/single through OUR restatement of Kaldi's fbank (Kaldi itself is absent: parity of the
    features is unpinned) and from there on through the reference's chain."""
    int2char = synth.phone_table_41()
    arpa = os.path.join(OUT, "lm_synth45.arpa")
    helpers = load_infer_functions()
    greedy = GreedyDecoder(int2char, space_idx=-1, blank_index=0)
    beam = BeamDecoder(int2char, beam_width=10, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)   # ctc_config.0329.yaml:87-88
    arrays, meta = {}, []
    for tag, g, B, T, L, seed in CHAIN_CASES:
        geom = synth.Geometry(**g)
        sd = synth.synth_state_dict(geom, seed=seed)
        x, x1, frac, tlen = synth.synth_batch(geom, B=B, T=T, L=L, seed=seed)
        m = build_reference_model(geom, sd)
        with torch.no_grad():
            logp = m(torch.from_numpy(x), torch.from_numpy(x1)).numpy()
        arrays[tag + "_logp"] = logp
        srt = np.sort(logp, axis=-1)
        recs = chain_records(logp, frac, x1, tlen, int2char, beam, greedy, helpers)
        meta.append(dict(tag=tag, geom=g, B=B, T=T, L=L, seed=seed, min_top2_gap=float((srt[..., -1] - srt[..., -2]).min()),
                         records=recs))
        print("G9", tag, logp.shape, "beam lens", [len(r["beam"].split()) for r in recs], "scores", [r["beam_chain"]["score"] for r in recs])
    # the WAV fixture: features by oracle.fbank + CMVN + stack/skip (ours), then the reference model and chain
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    sys.path.pop(0)
    import wave
    w = wave.open(os.path.join(OUT, "vocabulary_single_1.wav"))
    wav = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32)
    feats = orc.stack_skip(orc.apply_cmvn(orc.fbank(wav), orc.read_cmvn_stats(os.path.join(OUT, "global_fbank_cmvn.txt"))))
    geom = synth.Geometry(**synth.REFERENCE)
    sd = synth.synth_state_dict(geom, seed=11)
    canon = "v ow k ae b y ah l eh r iy"
    c2i = {v: k for k, v in int2char.items()}
    x1 = np.array([[c2i[p] for p in canon.split()]], dtype=np.int64)
    m = build_reference_model(geom, sd)
    with torch.no_grad():
        logp = m(torch.from_numpy(feats[None]), torch.from_numpy(x1)).numpy()
    arrays["wav_feats"] = feats
    arrays["wav_logp"] = logp
    recs = chain_records(logp, np.ones(1, dtype=np.float32), x1, np.array([x1.shape[1]]), int2char, beam, greedy, helpers)
    meta.append(dict(tag="wav", geom=dict(synth.REFERENCE), B=1, T=int(feats.shape[0]), L=int(x1.shape[1]), seed=11, canonical=canon,
                     records=recs))
    print("G9 wav", logp.shape, recs[0]["beam"], recs[0]["beam_chain"]["score"])
    np.savez_compressed(os.path.join(OUT, "g9_chain.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "g9_chain.json"), "w"))


# ----------------------------------------------------------------------------- G10 (augmentation + offline lexicon)
def gen_aug_goldens():
    """spec_augment / data_enhancement of AA/utils/tools.py under seeded generators (both numpy's and Python's global RNGs
    are consumed, in the reference's order), exactly as SpeechDataset.__getitem__ applies them (data_loader.py:132-137);
    the confusion table the type-2 enhancement draws from is written next to the mirror as data; and the CMU-dictionary
    lookup of AA/dict/phonetic_dict.py (load_cmudict / cmu_dict, extracted with ast: the module imports TTS / espeak
    packages that do not exist here) for the words of egs/vocabulary/single, with the stress post-processing of
    AA/infer.py:543-548."""
    import random
    import string
    json.dump(ref_tools.common_incorrect_voc, open(os.path.join(ROOT, "ctc-attention-mispronunciation_amd", "utils", "common_incorrect_voc.json"), "w"),
              indent=0, sort_keys=True)
    arrays, recs = {}, []
    rs = np.random.Generator(np.random.PCG64(4))
    for k, (T, D) in enumerate(((37, 81), (60, 81), (6, 81), (50, 40))):
        feat = rs.standard_normal((T, D)).astype(np.float32)             # kaldiio.load_mat hands float32 matrices
        trans = [int(v) for v in rs.integers(2, 44, size=int(rs.integers(3, 30)))]
        for seed in (0, 1, 2):
            random.seed(100 * k + seed); np.random.seed(100 * k + seed)
            f2 = ref_tools.spec_augment(feat)                                        # data_loader.py:134
            t2 = sum([ref_tools.data_enhancement(t) for t in trans], [])             # :136-137
            arrays["feat%d" % k] = feat
            arrays["aug%d_%d" % (k, seed)] = f2
            recs.append(dict(k=k, seed=seed, trans=trans, trans_aug=[int(v) for v in t2]))
    enh = []
    for etype in (1, 2, 3, 4):
        for prob in (0.1, 0.9):
            random.seed(7 * etype + int(prob * 10))
            seq = [int(v) for v in np.random.Generator(np.random.PCG64(etype)).integers(2, 44, size=60)]
            enh.append(dict(type=etype, prob=prob, seed=7 * etype + int(prob * 10), seq=seq,
                            out=[int(ref_tools.data_enhancement(p, prob, etype)[0]) for p in seq]))
    # the lexicon: reference methods run on a bare object
    src = open(os.path.join(AA, "dict", "phonetic_dict.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "Phonetic"][0]
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("load_cmudict", "cmu_dict")]
    stub = ast.ClassDef(name="RefLexicon", bases=[], keywords=[], body=meths, decorator_list=[])
    ns = {"os": os, "__file__": os.path.join(AA, "dict", "phonetic_dict.py")}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[stub], type_ignores=[])), "phonetic_dict.py<lexicon>", "exec"), ns)
    lex = ns["RefLexicon"]()
    lex.cmudict_plain = {}
    words = []
    vdir = "/root/reference/egs/vocabulary/single"
    for fn in sorted((f for f in os.listdir(vdir) if f.endswith(".txt")), key=lambda f: int(f.split(".")[0])):
        words.append(open(os.path.join(vdir, fn)).read().strip())
    words += ["The", "READ", "tomato", "zzzznotaword"]
    lexrec = []
    keep = set()
    for w_ in words:
        p1 = lex.cmu_dict(w_)
        model = None
        if p1:
            parts_ = [p.rstrip(string.digits) if p not in ["ER0", "AH0"] else p for p in p1.split(" ")]     # infer.py:545-547
            model = " ".join(p.lower() for p in parts_)
            keep.add(w_.lower())
        lexrec.append(dict(word=w_, cmu=p1, model=model))
    # data fixture: the dictionary lines of those words (and their alternates, which follow the reference's last-one-wins load)
    with open(os.path.join(AA, "dict", "cmudict.dict")) as f, open(os.path.join(OUT, "cmudict_subset.dict"), "w") as o:
        for line in f:
            key = line.split(" ")[0].strip().lower()
            if key in keep or key.split("(")[0] in keep:
                o.write(line)
    np.savez_compressed(os.path.join(OUT, "g10_aug.npz"), **arrays)
    json.dump(dict(items=recs, enhancement=enh, lexicon=lexrec, vowels=ref_tools.vowels, consonants=ref_tools.consonants),
              open(os.path.join(OUT, "g10_aug.json"), "w"))
    print("G10 aug: %d items, %d enhancement runs, lexicon:" % (len(recs), len(enh)), [(r["word"], r["model"]) for r in lexrec[:4]], "...")


# ----------------------------------------------------------------------------- G12 (BASELINE configs[0]: every word of egs/vocabulary/single)
def gen_words_goldens():
    """configs[0] over all 20 WAV / TXT pairs of egs/vocabulary/single (copied to tests/golden/vocabulary_single/ as data):
    TXT -> the reference's CMU-dictionary lookup (AA/dict/phonetic_dict.py load_cmudict / cmu_dict, extracted with ast as in G10) ->
    the stress post-processing of AA/infer.py:543-548 -> canonical ids; WAV -> OUR restatement of Kaldi's fbank + CMVN + stack/skip
    (Kaldi itself is absent: feature parity unpinned) -> the REFERENCE model (seeded synthetic weights: the trained checkpoint is not
    in the tree) -> the reference's chain (AA/infer.py:294-342: BeamDecoder(10) / GreedyDecoder, wer, align_canonical_decoded,
    stastics, score).  Words the dictionary does not hold (the reference then asks g2p_en / espeak, absent offline) are recorded
    without a chain."""
    import shutil
    import wave
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    sys.path.pop(0)
    int2char = synth.phone_table_41()
    c2i = {v: k for k, v in int2char.items()}
    arpa = os.path.join(OUT, "lm_synth45.arpa")
    helpers = load_infer_functions()
    greedy = GreedyDecoder(int2char, space_idx=-1, blank_index=0)
    beam = BeamDecoder(int2char, beam_width=10, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)
    src = open(os.path.join(AA, "dict", "phonetic_dict.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "Phonetic"][0]
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("load_cmudict", "cmu_dict")]
    stub = ast.ClassDef(name="RefLexicon", bases=[], keywords=[], body=meths, decorator_list=[])
    ns = {"os": os, "__file__": os.path.join(AA, "dict", "phonetic_dict.py")}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[stub], type_ignores=[])), "phonetic_dict.py<lexicon>", "exec"), ns)
    lex = ns["RefLexicon"]()
    lex.cmudict_plain = {}
    import string
    vdir = "/root/reference/egs/vocabulary/single"
    wdir = os.path.join(OUT, "vocabulary_single")
    os.makedirs(wdir, exist_ok=True)
    geom = synth.Geometry(**synth.REFERENCE)
    sd = synth.synth_state_dict(geom, seed=11)
    m = build_reference_model(geom, sd)
    stats = orc.read_cmvn_stats(os.path.join(OUT, "global_fbank_cmvn.txt"))
    arrays, meta = {}, []
    for i in range(1, 21):
        for ext in ("wav", "txt"):
            shutil.copyfile(os.path.join(vdir, "%d.%s" % (i, ext)), os.path.join(wdir, "%d.%s" % (i, ext)))
            os.chmod(os.path.join(wdir, "%d.%s" % (i, ext)), 0o644)
        word = open(os.path.join(vdir, "%d.txt" % i)).read().strip()
        w = wave.open(os.path.join(vdir, "%d.wav" % i))
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        wav = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32)
        cmu = lex.cmu_dict(word)
        rec = dict(i=i, word=word, cmu=cmu, samples=int(wav.size))
        if cmu:
            parts_ = [p.rstrip(string.digits) if p not in ["ER0", "AH0"] else p for p in cmu.split(" ")]      # infer.py:545-547
            canon = " ".join(p.lower() for p in parts_)
            feats = orc.stack_skip(orc.apply_cmvn(orc.fbank(wav), stats))
            x1 = np.array([[c2i[p] for p in canon.split()]], dtype=np.int64)
            with torch.no_grad():
                logp = m(torch.from_numpy(feats[None]), torch.from_numpy(x1)).numpy()
            arrays["logp%d" % i] = logp
            rec.update(canonical=canon, T=int(feats.shape[0]),
                       records=chain_records(logp, np.ones(1, dtype=np.float32), x1, np.array([x1.shape[1]]), int2char, beam, greedy, helpers))
            print("G12", i, word, "->", canon, "| beam:", rec["records"][0]["beam"], "| score", rec["records"][0]["beam_chain"].get("score"))
        else:
            rec.update(canonical=None)
            print("G12", i, word, "-> not in cmudict (no chain)")
        meta.append(rec)
    np.savez_compressed(os.path.join(OUT, "g12_words.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "g12_words.json"), "w"))


# ----------------------------------------------------------------------------- G11 (one training step)
class GivenDropout(nn.Module):
    """nn.Dropout with the mask handed in: y = x * (mask / (1-p)), the arithmetic of ATen's dropout with a fixed noise tensor."""

    def __init__(self, mask, p):
        super(GivenDropout, self).__init__()
        self.noise = torch.from_numpy(mask.astype(np.float32)) * (1.0 / (1.0 - p))

    def forward(self, x):
        return x * self.noise


def gen_train_goldens():
    """The REFERENCE model in train mode (BatchNorm on batch statistics, its Dropout modules replaced by GivenDropout with seeded
    masks), loss = nn.CTCLoss(sum)(out, targets, in_len, tgt_len) / B, loss.backward()  (AA/steps/train_ctc.py:63-76):
    log-probs, loss, every parameter gradient and the updated running statistics.  Tiny geometry in full; reference geometry
    (H=384, 4 layers; a short case and one with T' = 80, ragged lengths) as norms + 48 sampled entries per tensor (the tensors
    have 21 M elements)."""
    arrays, meta = {}, []
    # "long": reference geometry at T' = 80 with ragged input / label lengths (synth_batch's default) -- BPTT over 80 steps
    for tag, g, seed, B, T, L, Lt in (("tiny", dict(synth.TINY), 11, 3, 12, 4, 2), ("ref", dict(synth.REFERENCE), 21, 2, 16, 5, 3),
                                      ("long", dict(synth.REFERENCE), 31, 3, 160, 9, 7)):
        geom = synth.Geometry(**g)
        sd, x, x1, masks, tg, il, tl = synth.train_case(geom, seed, B, T, L, Lt)
        m = build_reference_model(geom, sd)
        m.train()
        drops = [m.conv[0], m.conv[1]] + [m.rnns[i] for i in range(geom.layers)]
        for mod, mk in zip(drops, masks):
            assert isinstance(mod.dropout, nn.Dropout) and abs(mod.dropout.p - 0.2) < 1e-12
            mod.dropout = GivenDropout(mk, 0.2)
        out = m(torch.from_numpy(x), torch.from_numpy(x1))
        loss = nn.CTCLoss(reduction="sum")(out, torch.from_numpy(tg), torch.from_numpy(il), torch.from_numpy(tl)) / B
        loss.backward()
        rec = dict(tag=tag, geom=g, seed=seed, B=B, T=T, L=L, Lt=Lt, loss=float(loss.item()), tensors={})
        arrays[tag + "_logp"] = out.detach().numpy()
        rs = np.random.Generator(np.random.PCG64(99))
        for k, prm in m.named_parameters():
            gr = prm.grad.numpy().ravel()
            if tag == "tiny" or gr.size <= 4096:
                arrays["%s_grad_%s" % (tag, k)] = prm.grad.numpy()
            else:
                idx_ = rs.integers(0, gr.size, size=48)
                arrays["%s_gidx_%s" % (tag, k)] = idx_
                arrays["%s_gval_%s" % (tag, k)] = gr[idx_]
            rec["tensors"][k] = dict(norm=float(np.sqrt((gr.astype(np.float64) ** 2).sum())), absmax=float(np.abs(gr).max()))
        for k, bufv in m.named_buffers():
            if "running_" in k:
                arrays["%s_run_%s" % (tag, k)] = bufv.numpy().copy()
        meta.append(rec)
        print("G11", tag, "loss %.6f" % rec["loss"], "grad norms fc.1 %.3e conv.0 %.3e" % (rec["tensors"]["fc.1.weight"]["norm"], rec["tensors"]["conv.0.conv.weight"]["norm"]))
    np.savez_compressed(os.path.join(OUT, "g11_train.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "g11_train.json"), "w"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["model", "decode", "align", "eval", "ctc", "input", "chain", "aug", "train", "words"]
    with torch.no_grad():
        if "model" in which:
            gen_model_goldens()
        if "decode" in which:
            gen_decode_goldens()
    if "align" in which:
        gen_align_goldens()
    if "eval" in which:
        gen_eval_goldens()
    if "ctc" in which:
        gen_ctc_goldens()
    if "input" in which:
        gen_input_goldens()
    if "chain" in which:
        gen_chain_goldens()
    if "aug" in which:
        gen_aug_goldens()
    if "train" in which:
        gen_train_goldens()
    if "words" in which:
        gen_words_goldens()
