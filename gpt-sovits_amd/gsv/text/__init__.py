"""Symbol tables and `cleaned_text_to_sequence` (reference GPT_SoVITS/text/__init__.py:14-28).

The v1 (322) and v2 (732) symbol inventories are data: `symbols_v1.json` / `symbols_v2.json` are written by
`oracle/gen_golden_text.py` from the reference's `text/symbols.py:399` and `text/symbols2.py:419` lists, and
`tests/test_text_frontend.py` pins the id mapping against the reference function's outputs.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name: str) -> dict:
    with open(os.path.join(_HERE, name), encoding="utf-8") as f:
        return json.load(f)


class _Table:
    def __init__(self, name):
        d = _load(name)
        self.symbols: List[str] = d["symbols"]
        self.punctuation: List[str] = d["punctuation"]
        self.pad: str = d["pad"]
        self.to_id: Dict[str, int] = {s: i for i, s in enumerate(self.symbols)}


_tables: Dict[str, _Table] = {}


def table(version: Optional[str] = None) -> _Table:
    """v1 -> the 322-symbol table, anything else (v2, v2Pro, v3, v4) -> the 732-symbol table (text/__init__.py:23-26)."""
    if version is None:
        version = os.environ.get("version", "v2")
    key = "v1" if version == "v1" else "v2"
    if key not in _tables:
        _tables[key] = _Table(f"symbols_{key}.json")
    return _tables[key]


def symbols(version: Optional[str] = None) -> List[str]:
    return table(version).symbols


def cleaned_text_to_sequence(cleaned_text, version: Optional[str] = None) -> List[int]:
    """phoneme symbols -> ids; an unknown symbol raises KeyError exactly like the reference (text/__init__.py:14-28)."""
    to_id = table(version).to_id
    return [to_id[s] for s in cleaned_text]
