"""Shared helpers for the test-suite (golden loading, geometry construction)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def record_margin(key, value, budget=None):
    """Keep a measured distance to a tolerance: merged into gpurun_out/margins.json (scratch on the GPU box; the round's copy is
    committed as profiles/round<N>_margins.json and cited in DESIGN.md section 2)."""
    path = os.path.join(ROOT, "gpurun_out", "margins.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    data[key] = {"measured": float(value)} if budget is None else {"measured": float(value), "budget": float(budget)}
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def jload(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def ids_to_beam_string(ids, int2char):
    return " ".join(int2char[i] for i in ids)


def ids_to_greedy_string(ids, int2char):
    return "".join(" " + int2char[i] for i in ids)


def chain_inputs(meta):
    """Regenerate the inputs of one G9 case (weights and batch come from the seed on both sides)."""
    from ctc_attention_mispronunciation_amd import synth
    geom = synth.Geometry(**meta["geom"])
    sd = synth.synth_state_dict(geom, seed=meta["seed"])
    if meta["tag"] == "wav":
        i2c = synth.phone_table_41()
        c2i = {v: k for k, v in i2c.items()}
        x1 = np.array([[c2i[p] for p in meta["canonical"].split()]], dtype=np.int64)
        return geom, sd, None, x1, np.ones(1, dtype=np.float32), np.array([x1.shape[1]])
    x, x1, frac, tlen = synth.synth_batch(geom, B=meta["B"], T=meta["T"], L=meta["L"], seed=meta["seed"])
    return geom, sd, x, x1, frac, tlen


def check_chain(records, beam_strings, greedy_strings, wer, diagnose):
    """The G9 records (made by the reference's own infer.py loop) against decoded strings from the path under test:
    strings identical, then wer -> align_canonical_decoded -> stastics -> score identical, through `wer(s1, s2)` and
    `diagnose(decoded, canonical)` of the implementation under test."""
    for b, rec in enumerate(records):
        assert beam_strings[b] == rec["beam"], ("beam", b)
        assert greedy_strings[b] == rec["greedy"], ("greedy", b)
        for name, hyp in (("beam", beam_strings[b]), ("greedy", greedy_strings[b])):
            want = rec[name + "_chain"]
            if "error" in want:
                try:
                    diagnose(hyp, rec["canonical"])
                except TypeError:
                    continue
                raise AssertionError("expected TypeError for an empty decode")
            got = diagnose(hyp, rec["canonical"])
            assert got["path"] == want["al_ops"] and got["decoded"] == want["al_hyp"] and got["canonical"] == want["al_can"], (name, b)
            assert (got["insertions"], got["substitutions"], got["deletions"]) == (want["ins"], want["sub"], want["dele"]), (name, b)
            assert (got["correct"], got["del_sub"], got["score"]) == (want["correct"], want["del_sub"], want["score"]), (name, b)
