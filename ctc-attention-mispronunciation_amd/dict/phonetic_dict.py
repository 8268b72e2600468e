"""Offline word -> canonical-phoneme lookup.

The reference's ``Phonetic`` (AA/dict/phonetic_dict.py) answers ``api_word_phones_cmu`` from an espeak phonemizer and
cross-checks it against g2p_en and the CMU dictionary; it also constructs a MeloTTS model at import.  None of those
packages exist offline.  This module keeps the one source that is a plain data file -- the CMU pronouncing dictionary the
reference ships as ``dict/cmudict.dict`` -- behind the same method names:

* ``load_cmudict`` / ``cmu_dict(word)``  (AA/dict/phonetic_dict.py:133-145, 443-454): the pronunciation stored under the
  lower-cased key, ARPAbet with stress digits, ``None`` for an unknown word;
* ``api_word_phones_cmu(word)``: here the dictionary answer (the reference returns the phonemizer's);
* ``phones_for_model(cmu_phones)``: the transformation ``infer.py`` applies before the ids go to the model
  (AA/infer.py:543-548): stress digits dropped except on ER0 / AH0 (separate classes of the 41-phone set), lower-cased.
"""
import os
import string

class Phonetic(object):
    def __init__(self, cmudict_path=None):
        self.cmudict_path = cmudict_path or os.environ.get("MDD_CMUDICT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmudict.dict")
        self.cmudict_plain = {}

    def load_cmudict(self, reload=False):
        if self.cmudict_plain and not reload:
            return
        if not os.path.exists(self.cmudict_path):
            raise FileNotFoundError("CMU dictionary not found at %s (pass cmudict_path= or set MDD_CMUDICT to the reference's "
                                    "dict/cmudict.dict)" % self.cmudict_path)
        table = {}
        with open(self.cmudict_path, "r") as f:
            for line in f:
                parts = [p.strip() for p in line.split(" ")]
                table[parts[0].lower()] = parts[1:]          # a later line for the same key replaces the earlier one, as in the reference
        self.cmudict_plain = table

    def cmu_dict(self, word, to_ipa=False):
        self.load_cmudict()
        phones = self.cmudict_plain.get(word.lower(), None)
        if not phones:
            return None
        if to_ipa:      # IPA rendering (stress-mark placement, AA/dict/phonetic_dict.py:367-400) is display code outside the path
            raise NotImplementedError("to_ipa=True: IPA display is not part of the offline lookup")
        return " ".join(phones)

    def api_word_phones_cmu(self, word):
        return self.cmu_dict(word.strip())

    @staticmethod
    def phones_for_model(cmu_phones):
        parts = [p.rstrip(string.digits) if p not in ("ER0", "AH0") else p for p in cmu_phones.split(" ")]
        return " ".join(p.lower() for p in parts)
