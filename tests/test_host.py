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
