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

special = [("￥", "zh", "SP2"), ("^", "zh", "SP3")]          # cleaner.py:13-18

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


def clean_text(text: str, language: str, version: Optional[str] = None) -> Tuple[List[str], Optional[List[int]], str]:
    if version is None:
        version = os.environ.get("version", "v2")
    key = "v1" if version == "v1" else "v2"
    symbols = table(key).to_id
    if language not in _LANGS[key]:            # cleaner.py:32-34
        language, text = "en", " "
    for special_s, special_l, target_symbol in special:
        if special_s in text and language == special_l:
            return clean_special(text, language, special_s, target_symbol, version)
    mod = _backend(language)
    norm_text = mod.text_normalize(text) if hasattr(mod, "text_normalize") else text
    if language in ("zh", "yue"):
        phones, word2ph = mod.g2p(norm_text)
        assert len(phones) == sum(word2ph)
        assert len(norm_text) == len(word2ph)
    elif language == "en":
        phones = list(mod.g2p(norm_text))
        if len(phones) < 4:
            phones = [","] + phones
        word2ph = None
    else:
        phones = list(mod.g2p(norm_text))
        word2ph = None
    phones = ["UNK" if ph not in symbols else ph for ph in phones]
    return phones, word2ph, norm_text


def clean_special(text: str, language: str, special_s: str, target_symbol: str, version: Optional[str] = None):
    """cleaner.py:58-83: the silence marks become "," for G2P and are mapped back to SP2 / SP3 afterwards."""
    key = "v1" if (version or os.environ.get("version", "v2")) == "v1" else "v2"
    symbols = table(key).to_id
    text = text.replace(special_s, ",")
    mod = _backend(language)
    norm_text = mod.text_normalize(text)
    phones = mod.g2p(norm_text)
    new_ph = []
    for ph in phones[0]:
        assert ph in symbols
        new_ph.append(target_symbol if ph == "," else ph)
    return new_ph, phones[1], norm_text
