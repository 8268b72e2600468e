"""CPU-side tests: the C-ABI library loads and exports every declared symbol, host-only entry points
(alignment, length bookkeeping), the LM reader and the diagnosis post-processing against goldens.
No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.helpers import npz, jload, GOLD, ROOT


def test_library_exports_every_declared_symbol():
    from ctc_attention_mispronunciation_amd import _lib
    header = open(os.path.join(ROOT, "include", "mdd_hip.h")).read()
    declared = set(re.findall(r"\b(mdd_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.lib().mdd_version() == 100


def test_alignment_goldens_through_decoder_wer():
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import Decoder
    dec = Decoder({0: "blank"}, space_idx=-1, blank_index=0)
    g = jload("g4_align.json")
    for r in g["pairs"]:
        dist, ops = dec.wer(" ".join(r["hyp"]), " ".join(r["can"]))
        assert dist == r["dist"] and ops == r["ops"], r
    for e in g["empties"]:
        with pytest.raises(TypeError):
            dec.wer(e["s1"], e["s2"])
    assert dec.cer("kitten", "sitting") == 3


def test_alignment_goldens_through_the_batch_entry_point():
    """mdd_align_batch: the golden pairs (reference wer + printChanges) as one padded batch, empty rows marked -1."""
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import align_ids_batch, _tokens_to_ids
    pairs = jload("g4_align.json")["pairs"]
    rows = [_tokens_to_ids(r["hyp"], r["can"]) for r in pairs] + [([], [1, 2]), ([3], [])]
    sa, sb = max(len(a) for a, _ in rows), max(len(b) for _, b in rows)
    A, Bm = np.full((len(rows), sa), -7, dtype=np.int32), np.full((len(rows), sb), -9, dtype=np.int32)   # padding never read
    for x, (a, b) in enumerate(rows):
        A[x, :len(a)] = a; Bm[x, :len(b)] = b
    dist, ops, nops = align_ids_batch(A, [len(a) for a, _ in rows], Bm, [len(b) for _, b in rows])
    for x, r in enumerate(pairs):
        assert dist[x] == r["dist"] and ["-SID"[o] for o in ops[x, :nops[x]]] == r["ops"], r
    assert list(dist[-2:]) == [-1, -1] and list(nops[-2:]) == [0, 0]
    d0, o0, n0 = align_ids_batch(np.zeros((0, 3), np.int32), [], np.zeros((0, 2), np.int32), [])
    assert d0.shape == (0,) and n0.shape == (0,)
    with pytest.raises(ValueError):
        align_ids_batch(A, [1], Bm, [1])


def test_diagnosis_goldens():
    from ctc_attention_mispronunciation_amd import infer_core as ic
    for r in jload("g4_align.json")["pairs"]:
        a1, a2, al = ic.align_canonical_decoded(list(r["hyp"]), list(r["can"]), list(r["ops"]))
        assert (a1, a2, al) == (r["al_hyp"], r["al_can"], r["al_ops"]), r
        ins, sub, dele = ic.stastics(al, a2, a1)
        assert (ins, sub, dele) == (r["ins"], r["sub"], r["dele"])
        if r["score"] is not None:
            score, ok, ds = ic.pronunciation_score(al, len(ins))
            assert (score, ok, ds) == (r["score"], r["correct"], r["del_sub"])
        assert list(ic.print_aligned_string(a1, a2, al)) == r["printed"]


def test_lm_reader_tables_match_reference_parser():
    from ctc_attention_mispronunciation_amd.utils.NgramLM import LanguageModel
    g = npz("g3_decode.npz")
    for s in jload("g3_decode.json")["sets"]:
        Cn, i2c = s["C"], dict(enumerate(s["int2char"]))
        for tag in ("", "_missing"):
            lm = LanguageModel(os.path.join(GOLD, "lm_synth%d%s.arpa" % (Cn, tag)))
            np.testing.assert_array_equal(lm.dense_table(i2c, Cn), g["lm%d%s" % (Cn, tag)])
        # 'blank' is no LM word (never looked up: prefixes hold no blank); every other pair resolves
        assert not np.isnan(g["lm%d" % Cn][1:, 1:]).any() and np.isnan(g["lm%d_missing" % Cn][1:, 1:]).any()


def test_length_bookkeeping_table():
    from ctc_attention_mispronunciation_amd import _lib
    import torch
    from ctc_attention_mispronunciation_amd.utils.data_loader import frames_from_fraction
    for ln, maxlen, tout, want in npz("g6_input.npz")["len_table"]:
        assert _lib.lib().mdd_len_frames(int(ln), int(maxlen), int(tout)) == want
        frac = torch.zeros(1)
        frac[0] = int(ln) / int(maxlen)
        assert int(frames_from_fraction(frac, int(tout))[0]) == want
    assert _lib.lib().mdd_stack_len(1000, 2, 2) == 500 and _lib.lib().mdd_stack_len(7, 2, 2) == 4


def test_collate_matches_reference_shapes():
    import torch
    from ctc_attention_mispronunciation_amd.utils.data_loader import create_input
    batch = [(torch.ones(6, 3), torch.tensor([2, 3]), torch.tensor([4, 5, 6]), "a"),
             (torch.ones(4, 3) * 2, torch.tensor([7]), torch.tensor([8]), "b")]
    data, isz, lab, lsz, tr, tsz, utts = create_input(batch)
    assert data.shape == (2, 6, 3) and float(data[1, 4:].abs().sum()) == 0
    assert isz.dtype == torch.float32 and float(isz[1]) == np.float32(4 / 6)
    assert lab.tolist() == [[2, 3], [7, 0]] and tr.tolist() == [[4, 5, 6], [8, 0, 0]] and utts == ["a", "b"]


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the product entry points must raise, never route through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ctc_attention_mispronunciation_amd import _lib
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import GreedyDecoder
    with pytest.raises(_lib.MddError):
        GreedyDecoder({0: "blank", 1: "a"}, space_idx=-1).decode(torch.zeros(3, 1, 2), [3])
    pkg = os.path.join(ROOT, "ctc-attention-mispronunciation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_augmentation_goldens_reproduce_the_reference_rng_stream():
    """spec_augment + data_enhancement (AA/utils/tools.py:229-255,290-359) as the data loader applies them
    to a training item (data_loader.py:132-137).  Both global generators (numpy's for mask widths, Python's for positions
    and mutations) are consumed in the reference's order, so seeded runs give the reference's outputs exactly (G10)."""
    import random
    from ctc_attention_mispronunciation_amd.utils import tools
    g, meta = npz("g10_aug.npz"), jload("g10_aug.json")
    assert tools.vowels == meta["vowels"] and tools.consonants == meta["consonants"]
    for r in meta["items"]:
        feat = g["feat%d" % r["k"]]
        random.seed(100 * r["k"] + r["seed"]); np.random.seed(100 * r["k"] + r["seed"])
        f2, t2 = tools.augment_item(feat, r["trans"], train=True)
        np.testing.assert_array_equal(f2, g["aug%d_%d" % (r["k"], r["seed"])])
        assert f2.dtype == feat.dtype and t2 == r["trans_aug"]
        f3, t3 = tools.augment_item(feat, r["trans"], train=False)
        assert f3 is feat and t3 == r["trans"]
    for e in meta["enhancement"]:
        random.seed(e["seed"])
        assert [tools.data_enhancement(p, e["prob"], e["type"])[0] for p in e["seq"]] == e["out"], (e["type"], e["prob"])
    # the host forms of stack / skip agree with the GPU path's golden (G6)
    g6 = npz("g6_input.npz")
    for i in range(6):
        st = tools.skip_feat(tools.make_context(g6["raw%d" % i], 0, 2), 2)
        if st.shape[0] % 2:
            st = np.vstack([st, np.zeros((1, st.shape[1]), dtype=st.dtype)])
        np.testing.assert_array_equal(st.astype(np.float32), g6["stk%d" % i])


def test_offline_lexicon_matches_reference_cmudict_lookup():
    """Word -> canonical phonemes without espeak / g2p_en / MeloTTS: the reference's own load_cmudict / cmu_dict answers
    for the words of egs/vocabulary/single (G10), then the stress handling of infer.py:543-548."""
    from ctc_attention_mispronunciation_amd.dict.phonetic_dict import Phonetic
    ph = Phonetic(os.path.join(GOLD, "cmudict_subset.dict"))
    for r in jload("g10_aug.json")["lexicon"]:
        assert ph.cmu_dict(r["word"]) == r["cmu"], r
        assert ph.api_word_phones_cmu(r["word"] + "\n") == r["cmu"]
        if r["cmu"]:
            assert Phonetic.phones_for_model(r["cmu"]) == r["model"]
    assert ph.cmu_dict("zzzznotaword") is None
    with pytest.raises(FileNotFoundError):
        Phonetic("/nonexistent/cmudict.dict").cmu_dict("a")


def test_speech_dataset_reads_ark_and_collates(tmp_path):
    """SpeechDataset / SpeechDataLoader over a Kaldi ark+scp written by the package's own writer: eval items are the stacked
    rows of the G6 golden arithmetic, train items are augmented, batches come out of create_input."""
    import random
    import types
    import torch
    from ctc_attention_mispronunciation_amd.utils import data_loader as dl, fbank as fb, tools
    rs = np.random.Generator(np.random.PCG64(2))
    feats = {"u%d" % i: rs.standard_normal((9 + 4 * i, 81)).astype(np.float32) for i in range(3)}
    # fbank's writer needs no GPU; it writes the scp index with byte offsets
    fb.write_ark_scp(str(tmp_path / "f.ark"), str(tmp_path / "f.scp"), feats)
    (tmp_path / "units").write_text("\n".join("%s %s" % (p, p) for p in "sil aa b k".split()) + "\n")
    (tmp_path / "lab").write_text("u0 aa b\nu1 k zz aa\nu2 b\n")
    (tmp_path / "trn").write_text("u0 aa k\nu1 k b aa\nu2 sil b\n")
    vocab = dl.Vocab(str(tmp_path / "units"))
    opts = types.SimpleNamespace(left_ctx=0, right_ctx=2, n_skip_frame=2, n_downsample=2, feature_type="fbank", mel=False)
    ds = dl.SpeechDataset(vocab, str(tmp_path / "f.scp"), str(tmp_path / "lab"), str(tmp_path / "trn"), opts, train=False)
    assert len(ds) == 3
    x, lab, tr, utt = ds[1]
    want = tools.skip_feat(tools.make_context(feats["u1"], 0, 2), 2)
    assert utt == "u1" and x.shape == (8, 243) and lab.tolist() == [vocab.word2index["k"], 1, vocab.word2index["aa"]]
    np.testing.assert_array_equal(x.numpy()[:7], want)
    assert not x.numpy()[7].any()                                   # 13 frames -> 7 kept -> padded to 8
    batch = next(iter(dl.SpeechDataLoader(ds, batch_size=3, shuffle=False)))
    assert batch[0].shape == (3, 10, 243) and batch[0].dtype == torch.float32 and batch[6] == ["u0", "u1", "u2"]
    np.testing.assert_allclose(batch[1].numpy(), np.array([6, 8, 10], dtype=np.float32) / np.float32(10))
    tr_ds = dl.SpeechDataset(vocab, str(tmp_path / "f.scp"), str(tmp_path / "lab"), str(tmp_path / "trn"), opts, train=True)
    random.seed(3); np.random.seed(3)
    xa = tr_ds[2][0]
    random.seed(3); np.random.seed(3)
    f2, _ = tools.augment_item(feats["u2"], [2, 4], train=True)
    np.testing.assert_array_equal(xa.numpy()[:9], tools.skip_feat(tools.make_context(f2, 0, 2), 2))


def test_lr_schedule_state_machine():
    """The dev-loss driven halving of train_ctc.py:207-268 (inline code in the reference's main(); restated, no function to call):
    patience of ten epochs inside the band, immediate decay outside it, best state restored, stop after eight adjustments."""
    import torch
    from ctc_attention_mispronunciation_amd.steps.train_ctc import LrSchedule
    net = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(net.parameters(), lr=1.0)
    s = LrSchedule(net, opt, init_lr=1.0, decay=0.5, end_adjust_acc=2.0, num_epoches=100)
    assert s.begin_epoch()
    s.end_epoch(0.5, 50.0)                      # first epoch: new best
    assert s.loss_best == 50.0 and s.adjust_rate_count == 0
    with torch.no_grad():
        net.weight.fill_(7.0)
    assert s.begin_epoch()
    s.end_epoch(0.4, 60.0)                      # far worse than best + band: decay at once, best weights restored
    assert s.adjust_time == 1 and s.adjust_rate_flag and float(net.weight[0, 0]) != 7.0
    assert s.begin_epoch() and opt.param_groups[0]["lr"] == 0.5 and s.learning_rate == 0.5
    for k in range(9):                          # inside the band: patience counts up
        s.end_epoch(0.4, 49.5)
        assert s.adjust_time == 1 and s.adjust_rate_count == k + 1, k
        assert s.begin_epoch()
    assert s.loss_best == 50.0 and s.loss_best_true == 49.5
    s.end_epoch(0.4, 49.0)                      # tenth epoch inside the band: decay, loss_best follows the true best
    assert s.adjust_time == 2 and s.loss_best == 49.0
    epochs = 0
    while s.begin_epoch():
        s.end_epoch(0.1, 99.0)
        epochs += 1
    assert s.adjust_time == 8 and s.stop_train and epochs == 6 and abs(s.learning_rate - 0.5 ** 7) < 1e-12
    # reference quirk kept: optimizer.load_state_dict(op_state) also restores the RATE stored with the best state, so the optimizer's
    # own lr is (rate at the best epoch) x decay, not the printed `learning_rate` (train_ctc.py:219-224, 262)
    assert opt.param_groups[0]["lr"] == 0.5
    assert s.acc_best == 0.5
    s.finish()


def test_host_entry_points_under_sanitizers(tmp_path):
    """The host-side C++ of the ABI (mdd_align, mdd_align_batch, mdd_eval_batch, length bookkeeping: csrc/host_align.cpp) built
    with -fsanitize=address,undefined and driven, in a child interpreter with the ASan runtime preloaded, through the alignment
    goldens (G4), the evaluation goldens (G8), edge cases (single tokens, stride-exact rows, empty rows) and random fuzz.  Any
    out-of-bounds access or undefined behaviour aborts the child."""
    import subprocess
    import sys
    src = os.path.join(ROOT, "ctc-attention-mispronunciation_amd", "csrc", "host_align.cpp")
    lib = str(tmp_path / "libmdd_host_asan.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-DMDD_HOST_STANDALONE", src, "-o", lib])
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    child = r'''
import ctypes as C, json, sys
import numpy as np
lib = C.CDLL(sys.argv[1])
gold = sys.argv[2]
vp = C.c_void_p
def p(a): return a.ctypes.data_as(vp)
def align(a, b):
    a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
    ops = np.zeros(len(a) + len(b), dtype=np.uint8)          # exactly the documented capacity: an overrun is an ASan error
    d, n = C.c_int32(0), C.c_int32(0)
    rc = lib.mdd_align(p(a), len(a), p(b), len(b), C.byref(d), p(ops), C.byref(n))
    return rc, d.value, ["-SID"[o] for o in ops[:n.value]]
g4 = json.load(open(gold + "/g4_align.json"))
for r in g4["pairs"]:
    voc = {}
    ia = [voc.setdefault(t, len(voc)) for t in r["hyp"]]; ib = [voc.setdefault(t, len(voc)) for t in r["can"]]
    rc, d, ops = align(ia, ib)
    assert rc == 0 and d == r["dist"] and ops == r["ops"], r
assert align([], [1])[0] == -5 and align([1], [])[0] == -5
rs = np.random.Generator(np.random.PCG64(1))
for _ in range(300):
    na, nb = int(rs.integers(1, 60)), int(rs.integers(1, 60))
    rc, d, ops = align(rs.integers(0, 5, na), rs.integers(0, 5, nb))
    assert rc == 0 and len(ops) <= na + nb and ops.count("S") + ops.count("I") + ops.count("D") == d
# batch entry point: rows exactly as wide as their pitch, empty rows, pitch check
for n in (0, 1, 7):
    sa, sb = 9, 11
    A = rs.integers(0, 4, (n, sa)).astype(np.int32); Bm = rs.integers(0, 4, (n, sb)).astype(np.int32)
    la = rs.integers(0, sa + 1, n).astype(np.int32); lb = rs.integers(0, sb + 1, n).astype(np.int32)
    dist = np.zeros(n, np.int32); nops = np.zeros(n, np.int32); ops = np.zeros((n, sa + sb), np.uint8)
    assert lib.mdd_align_batch(p(A), p(la), sa, p(Bm), p(lb), sb, n, p(dist), p(ops), sa + sb, p(nops)) == 0
    for x in range(n):
        if la[x] == 0 or lb[x] == 0: assert dist[x] == -1 and nops[x] == 0
        else: assert (dist[x], list(ops[x, :nops[x]])) == (align(A[x, :la[x]], Bm[x, :lb[x]])[1], ["-SID".index(o) for o in align(A[x, :la[x]], Bm[x, :lb[x]])[2]])
    if n:
        assert lib.mdd_align_batch(p(A), p(la), sa, p(Bm), p(lb), sb, n, p(dist), p(ops), 3, p(nops)) in (0, -1)   # too small a pitch is refused (or no row needs it)
# evaluation counts against G8
g8 = json.load(open(gold + "/g8_eval.json"))
keys = ("total", "TA", "FR", "FA", "TRc", "TRw", "total_wer", "num_word")
for b in g8["batches"]:
    voc = {}
    rows = [[[voc.setdefault(t, len(voc)) for t in s.split(" ") if t] for s in b[k]] for k in ("decoded_nosil", "labels_nosil", "canonicals_nosil")]
    n = len(rows[0]); stride = max(len(r) for grp in rows for r in grp)
    mats, lens = [], []
    for grp in rows:
        m = np.full((n, stride), -1, np.int32); l = np.zeros(n, np.int32)
        for x, r in enumerate(grp): m[x, :len(r)] = r; l[x] = len(r)
        mats.append(m); lens.append(l)
    counts = np.zeros(8, np.int64)
    rc = lib.mdd_eval_batch(p(mats[0]), p(lens[0]), p(mats[1]), p(lens[1]), p(mats[2]), p(lens[2]), n, stride, p(counts))
    assert rc == 0 and list(counts) == [b[k] for k in keys], (list(counts), [b[k] for k in keys])
z = np.zeros(1, np.int32); one = np.ones((1, 1), np.int32); c8 = np.zeros(8, np.int64)
assert lib.mdd_eval_batch(p(one), p(z), p(one), p(z + 1), p(one), p(z + 1), 1, 1, p(c8)) == -5     # empty decode: MDD_ERR_EMPTY
lib.mdd_stack_len.restype = C.c_int32; lib.mdd_len_frames.restype = C.c_int32
assert lib.mdd_stack_len(1000, 2, 2) == 500 and lib.mdd_stack_len(1, 2, 2) == 2 and lib.mdd_len_frames(3, 7, 250) >= 0
print("sanitized host entry points ok")
'''
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", child, lib, GOLD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitized host entry points ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_asm_mfma_hazards(tmp_path):
    """The kernels that issue MFMAs from inline asm (the persistent BiLSTM layers, the f32x6 GEMM) hide those instructions' register reads
    from the compiler's hazard recognizer.  The gfx950 ISA of those translation units, rebuilt here, must be free of the two patterns that
    produced silently wrong numbers (profiles/round3_lstm_ordering.txt): a vector instruction writing an MFMA operand fewer than two wait
    states in front of the MFMA, and dependent MFMAs issued back to back (tools/check_mfma_hazards.py)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    csrc = os.path.join(ROOT, "ctc-attention-mispronunciation_amd", "csrc")
    procs = []
    for unit, pat in (("lstm", ["granule"]), ("lstm_f32", ["layer_f32"]), ("gemm_bf16x6", ["f32x6"])):
        out = str(tmp_path / (unit + ".s"))
        procs.append((unit, pat, out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result",
                                                        "-Wno-inline-asm", "--offload-device-only", "-S", "-o", out, os.path.join(csrc, unit + ".hip")],
                                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for unit, pat, out, pr in procs:
        log = pr.communicate(timeout=900)[0]
        assert pr.returncode == 0, log[-2000:]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"), out] + pat, capture_output=True, text=True)
        assert r.returncode == 0 and " 0 finding(s)" in r.stdout, unit + ":\n" + r.stdout[:3000]
        assert not r.stdout.startswith("0 kernel(s)"), unit
