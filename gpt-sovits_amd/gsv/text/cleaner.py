"""`clean_text` (reference GPT_SoVITS/text/cleaner.py:21-55): text -> (phoneme symbols, word2ph, normalised text).

The reference dispatches to one G2P module per language (`text/chinese2.py`, `japanese.py`, `english.py`, `korean.py`,
`cantonese.py`), all of which sit on third-party packages (pypinyin, g2pw, pyopenjtalk, g2p_en, nltk, jamo, ...) that are
outside the hot-path scope ("G2P libs stay third-party").  Here the per-language module is a
PLUG-IN with the same two-function interface (`text_normalize(text) -> str`, `g2p(norm_text) -> phones` or
`(phones, word2ph)` for zh / yue); everything around it -- language map per version, the unknown-language fallback, the
"fewer than 4 English phones -> leading comma" rule, the SP2 / SP3 special marks, the UNK replacement -- is the reference's.

Built-in back-ends (gsv/text/g2p.py): `SymbolG2P` (the text already is a phoneme string) and `DictG2P` (English, CMUdict-format
dictionary supplied by the user).  `register_g2p("zh", module)` installs any object with the interface above.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

from . import table

# silence marks (reference cleaner.py:13-18): (mark, language it applies to, symbol it becomes)
special = [("￥", "zh", "SP2"), ("^", "zh", "SP3")]

_LANGS = {"v1": ("zh", "ja", "en"), "v2": ("zh", "ja", "en", "ko", "yue")}      # cleaner.py:26-30
_registry: Dict[str, object] = {}


def register_g2p(language: str, backend) -> None:
    """backend: object (or module) with g2p(norm_text) and optionally text_normalize(text)."""
    _registry[language] = backend


def registered() -> List[str]:
    return sorted(_registry)


def _backend(language: str):
    if language not in _registry:
        raise NotImplementedError(
            f"no G2P back-end registered for language '{language}': install one with gsv.text.cleaner.register_g2p "
            f"(the reference's text/*.py modules sit on third-party G2P packages, outside the hot-path scope)")
    return _registry[language]


def _table_key(version: Optional[str]) -> str:
    return "v1" if (version or os.environ.get("version", "v2")) == "v1" else "v2"


# What a language's G2P returns and how its phones are finished, as data (reference cleaner.py:36-52 spells it as an if-chain):
#   aligned  -- g2p returns (phones, word2ph) and both must line up with the normalised text (zh, yue)
#   min_len  -- fewer phones than this get a leading "," (en: 4)
_SHAPE = {"zh": {"aligned": True}, "yue": {"aligned": True}, "en": {"min_len": 4}}


def _run_g2p(language: str, text: str):
    """-> (phones, word2ph or None, normalised text) of one language's back-end"""
    mod = _backend(language)
    norm = mod.text_normalize(text) if hasattr(mod, "text_normalize") else text
    shape = _SHAPE.get(language, {})
    out = mod.g2p(norm)
    if shape.get("aligned"):
        phones, word2ph = out
        assert len(phones) == sum(word2ph) and len(norm) == len(word2ph)
        return list(phones), word2ph, norm
    phones = list(out)
    if len(phones) < shape.get("min_len", 0):
        phones.insert(0, ",")
    return phones, None, norm


def clean_text(text: str, language: str, version: Optional[str] = None) -> Tuple[List[str], Optional[List[int]], str]:
    key = _table_key(version)
    known = table(key).to_id
    if language not in _LANGS[key]:            # an unknown language is synthesised as one English blank (cleaner.py:32-34)
        language, text = "en", " "
    mark = next(((m, sym) for m, lang, sym in special if lang == language and m in text), None)
    if mark is not None:
        return clean_special(text, language, mark[0], mark[1], version)
    phones, word2ph, norm = _run_g2p(language, text)
    return [ph if ph in known else "UNK" for ph in phones], word2ph, norm


def clean_special(text: str, language: str, special_s: str, target_symbol: str, version: Optional[str] = None):
    """cleaner.py:58-83: the silence marks become "," for G2P and are mapped back to SP2 / SP3 afterwards."""
    known = table(_table_key(version)).to_id
    mod = _backend(language)
    norm = mod.text_normalize(text.replace(special_s, ","))
    phones, word2ph = mod.g2p(norm)
    assert all(ph in known for ph in phones)
    return [target_symbol if ph == "," else ph for ph in phones], word2ph, norm
