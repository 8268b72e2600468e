"""Per-utterance mispronunciation diagnosis -- the host post-processing of AA/infer.py that turns a
decoded phoneme string and the canonical phoneme string into the printed diagnosis and score.

Reference: AA/infer.py:155-209 (print_aligned_string, align_canonical_decoded), :405-433 (stastics),
:304-342 (sil removal, 'err' stripping, score).  Pure Python on <= ~50 tokens per utterance; function
names are the reference's so callers can switch by changing the import.
"""
import math


def print_aligned_string(s1, s2, l):
    pad = lambda p: p + " " if len(p) == 1 else p   # noqa: E731
    return " ".join(pad(p) for p in s1), " ".join(pad(p) for p in s2), " ".join(s + " " for s in l)


def align_canonical_decoded(s1, s2, l):
    """s1 decoded phones, s2 canonical phones, l the op path of Decoder.wer(decoded, canonical).
    Returns the two sequences padded to the length of the path ('D' placeholder in the decoded row where
    a canonical phone was deleted, 'I' placeholder in the canonical row under an inserted phone), then
    (a) all but one of a run of leading insertions dropped, (b) a leading insertion dropped when it
    merely repeats the first aligned decoded phone."""
    hyp, can, ops = [], [], list(l)
    di = ci = 0
    lead = 0
    for pos, op in enumerate(ops):
        if op == "-" or op == "S":
            hyp.append(s1[di]); can.append(s2[ci]); di += 1; ci += 1
        elif op == "D":
            hyp.append("D"); can.append(s2[ci]); ci += 1
        else:
            hyp.append(s1[di]); can.append("I"); di += 1
            if lead == pos:
                lead += 1
    if lead > 0:
        hyp, can, ops = hyp[lead - 1:], can[lead - 1:], ops[lead - 1:]
    if ops[0] == "I" and can[0] == "I" and len(hyp) >= 2 and hyp[0] == hyp[1]:
        hyp, can, ops = hyp[1:], can[1:], ops[1:]
    return hyp, can, ops


def stastics(dc_path, phones_canonicals, phones_decoded):
    """(insertions, substitutions, deletions): the decoded phone under every 'I', the canonical phone over
    every 'S' and every 'D' (all three rows are index-aligned after align_canonical_decoded)."""
    ins = [phones_decoded[i] for i, op in enumerate(dc_path) if op == "I"]
    sub = [phones_canonicals[i] for i, op in enumerate(dc_path) if op == "S"]
    dele = [phones_canonicals[i] for i, op in enumerate(dc_path) if op not in "-SI"]
    return ins, sub, dele


def pronunciation_score(dc_path, n_insertions):
    """infer.py:338-342: ceil((1 - (DS + min(#ins/4, 0.1*(C+DS))) / (DS + C)) * 100)."""
    ds = sum(1 for c in dc_path if c == "D" or c == "S")
    ok = sum(1 for c in dc_path if c == "-")
    penalty = min(n_insertions / 4, 0.1 * (ok + ds))
    return math.ceil((1 - (ds + penalty) / (ds + ok)) * 100), ok, ds


def strip_sil(phones):
    return [p for p in phones if p != "sil"]


def diagnose(decoded, canonical, decoder, to_display=None):
    """One utterance of the loop at infer.py:304-342.  decoded / canonical: space-separated phoneme strings
    (as the decoders return them).  Returns a dict with the aligned rows, fault lists and score."""
    hyp = " ".join(strip_sil(decoded.split(" ")))
    ref = " ".join(strip_sil(canonical.split(" ")))
    hyp = hyp.replace("err", "").replace("  ", " ")
    _, path = decoder.wer(hyp, ref)
    ph_dec = [c for c in hyp.split(" ") if c]
    ph_can = [c for c in ref.split(" ") if c]
    if to_display is not None:
        ph_dec = [to_display.get(c.upper(), c) for c in ph_dec]
        ph_can = [to_display.get(c.upper(), c) for c in ph_can]
    ph_dec, ph_can, path = align_canonical_decoded(ph_dec, ph_can, path)
    ins, sub, dele = stastics(path, ph_can, ph_dec)
    score, ok, ds = pronunciation_score(path, len(ins))
    return dict(decoded=ph_dec, canonical=ph_can, path=path, insertions=ins, substitutions=sub, deletions=dele,
                correct=ok, del_sub=ds, score=score, printed=print_aligned_string(ph_dec, ph_can, path))
