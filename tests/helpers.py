"""Shared helpers for the test-suite (golden loading, geometry construction)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def jload(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def ids_to_beam_string(ids, int2char):
    return " ".join(int2char[i] for i in ids)


def ids_to_greedy_string(ids, int2char):
    return "".join(" " + int2char[i] for i in ids)
