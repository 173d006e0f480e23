"""Built-in G2P back-ends for `gsv.text.cleaner` (the reference's are third-party wrappers, out of scope).

* `SymbolG2P`   -- the "text" already is a phoneme string: symbols separated by white space ("HH AH0 L OW1 , W ER1 L D .").
                   Lets the CLI and the tests drive the whole pipeline without any G2P package.
* `DictG2P`     -- English by dictionary look-up in a CMUdict-format file the user supplies (the reference ships
                   text/cmudict.rep and cmudict-fast.rep in this format).  Follows the reference's look-up order for words the
                   dictionary knows (reference text/english.py:270-361: single letters, "A" -> EY1, first pronunciation,
                   possessive 's by the last phoneme's voicing, out-of-vocabulary words of <= 3 letters spelled out); words
                   the dictionary does not know are spelled out letter by letter -- the reference asks g2p_en's neural
                   predictor, nltk's tagger and wordsegment, none of which exist here.  PARITY UNPINNED.
"""
from __future__ import annotations

import re
from typing import Dict, List

_REP = {"；": ",", ";": ",", "：": ",", ":": ",", "，": ",", "。": ".", "！": "!", "？": "?", "’": "'", '"': "'"}   # english.py:26-32
_PUNCT = set("!?…,.-")
_ONES = "zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen " \
        "eighteen nineteen".split()
_TENS = "_ _ twenty thirty forty fifty sixty seventy eighty ninety".split()


def number_to_words(n: int) -> str:
    if n < 20:
        return _ONES[n]
    if n < 100:
        return _TENS[n // 10] + ("" if n % 10 == 0 else " " + _ONES[n % 10])
    if n < 1000:
        return _ONES[n // 100] + " hundred" + ("" if n % 100 == 0 else " " + number_to_words(n % 100))
    if n < 1000000:
        return number_to_words(n // 1000) + " thousand" + ("" if n % 1000 == 0 else " " + number_to_words(n % 1000))
    return " ".join(_ONES[int(c)] for c in str(n))


def replace_consecutive_punctuation(text: str) -> str:
    p = "".join(re.escape(c) for c in _PUNCT)
    return re.sub(f"([{p}])([{p}])+", r"\1", text)


class SymbolG2P:
    def text_normalize(self, text: str) -> str:
        return " ".join(text.split())

    def g2p(self, norm_text: str) -> List[str]:
        return norm_text.split()


class DictG2P:
    def __init__(self, path: str):
        self.cmu: Dict[str, List[str]] = {}
        with open(path, encoding="utf-8", errors="replace") as f:
            for line in f:
                if line.startswith(";;;") or not line.strip():
                    continue
                word, *ph = line.split()
                if word.endswith(")") and "(" in word:      # alternate pronunciations WORD(1): the reference keeps the first
                    continue
                self.cmu.setdefault(word.lower(), ph)
        if not self.cmu:
            raise ValueError(f"{path}: no CMUdict entries found")

    def text_normalize(self, text: str) -> str:
        text = "".join(_REP.get(c, c) for c in text)
        text = re.sub(r"\d+", lambda m: " " + number_to_words(int(m.group())) + " ", text)
        return replace_consecutive_punctuation(re.sub(r" {2,}", " ", text).strip())

    def _letters(self, word: str) -> List[str]:
        out: List[str] = []
        for w in word:
            if w == "a":
                out.append("EY1")
            elif not w.isalpha():
                out.append(w)
            else:
                out.extend(self.cmu.get(w, ["UNK"]))
        return out

    def qryword(self, o_word: str) -> List[str]:
        word = o_word.lower()
        if len(word) > 1 and word in self.cmu:
            return list(self.cmu[word])
        if len(word) <= 3:
            return self._letters(word)
        m = re.match(r"^([a-z]+)('s)$", word)
        if m:
            ph = self.qryword(m.group(1))
            if ph and ph[-1] in ("P", "T", "K", "F", "TH", "HH"):
                return ph + ["S"]
            if ph and ph[-1] in ("S", "Z", "SH", "ZH", "CH", "JH"):
                return ph + ["AH0", "Z"]
            return ph + ["Z"]
        return self._letters(word)

    def g2p(self, norm_text: str) -> List[str]:
        phones: List[str] = []
        for tok in re.findall(r"[A-Za-z']+|[^\sA-Za-z']", norm_text):
            low = tok.lower()
            if re.search("[a-z]", low) is None:
                if tok in _PUNCT:
                    phones.append(tok)
            elif len(low) == 1:
                phones.extend(["EY1"] if tok == "A" else self.cmu.get(low, ["UNK"]))
            else:
                phones.extend(self.qryword(tok))
        return phones
