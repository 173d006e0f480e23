"""Sentence cutting methods of the text front-end (N1: reference GPT_SoVITS/TTS_infer_pack/text_segmentation_method.py).

Same registry and method names (`cut0` .. `cut5`, `get_method`, `get_method_names`, `split`, `split_big_text`, `splits`,
`punctuation`) and the same outputs -- tests/test_host_logic.py checks them against outputs of the reference module --
but written here as small regex / generator based routines.  Pure host string work: the step that defines utterance
boundaries in front of the hot path.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, Iterator, List

punctuation = frozenset("!?…,.- ")
splits = frozenset("，。？！,.?!~:：—…")
_SPLIT_CLASS = "[" + re.escape("".join(sorted(splits))) + "]"
_CUT5_MARKS = frozenset(",.;?!、，。？！;：…")

METHODS: Dict[str, Callable[[str], str]] = {}


def register_method(name: str):
    def deco(fn):
        METHODS[name] = fn
        return fn
    return deco


def get_method(name: str) -> Callable[[str], str]:
    try:
        return METHODS[name]
    except KeyError:
        raise ValueError(f"Method {name} not found") from None


def get_method_names() -> list:
    return list(METHODS)


def _only_punct(s: str, marks=punctuation) -> bool:
    return all(ch in marks for ch in s)


def _keep_spoken(pieces, marks=punctuation) -> str:
    return "\n".join(p for p in pieces if not _only_punct(p, marks))


def split_big_text(text: str, max_len: int = 510) -> List[str]:
    """greedy packing of punctuation-delimited pieces into chunks of at most max_len characters (:42-68)"""
    out: List[str] = []
    cur = ""
    for piece in re.split("(" + _SPLIT_CLASS + ")", text):
        if len(cur) + len(piece) > max_len:
            out.append(cur)
            cur = piece
        else:
            cur += piece
    if cur:
        out.append(cur)
    return out


def _sentences(text: str) -> Iterator[str]:
    start = 0
    for m in re.finditer(_SPLIT_CLASS, text):
        yield text[start:m.end()]
        start = m.end()


def split(todo_text: str) -> List[str]:
    """sentences each ending with one split mark; a trailing mark is appended when missing (:71-88)"""
    t = todo_text.replace("……", "。").replace("——", "，")
    if t[-1] not in splits:
        t += "。"
    return list(_sentences(t))


@register_method("cut0")
def cut0(inp: str) -> str:                       # no cutting
    return "/n" if _only_punct(inp) else inp


@register_method("cut1")
def cut1(inp: str) -> str:                       # groups of four sentences
    inp = inp.strip("\n")
    sents = split(inp)
    if len(sents) <= 4:
        groups = [inp]
    else:
        starts = list(range(0, len(sents), 4))
        # the reference turns the LAST start index into the end marker, so the final group of <= 4 sentences is merged
        # into the one before it (split_idx[-1] = None, :108-113)
        bounds = starts[:-1] + [len(sents)]
        groups = ["".join(sents[bounds[i]:bounds[i + 1]]) for i in range(len(bounds) - 1)]
    return _keep_spoken(groups)


@register_method("cut2")
def cut2(inp: str) -> str:                       # about 50 characters per piece
    inp = inp.strip("\n")
    sents = split(inp)
    if len(sents) < 2:
        return inp
    groups: List[str] = []
    buf, n = "", 0
    for s in sents:
        buf += s
        n += len(s)
        if n > 50:
            groups.append(buf)
            buf, n = "", 0
    if buf:
        groups.append(buf)
    if len(groups) > 1 and len(groups[-1]) < 50:   # a short tail joins its predecessor
        tail = groups.pop()
        groups[-1] += tail
    return _keep_spoken(groups)


@register_method("cut3")
def cut3(inp: str) -> str:                       # at the Chinese full stop
    return _keep_spoken(inp.strip("\n").strip("。").split("。"))


@register_method("cut4")
def cut4(inp: str) -> str:                       # at the English full stop, not inside numbers
    return _keep_spoken(re.split(r"(?<!\d)\.(?!\d)", inp.strip("\n").strip(".")))


@register_method("cut5")
def cut5(inp: str) -> str:                       # at every punctuation mark, decimal points excepted
    inp = inp.strip("\n")
    pieces: List[str] = []
    start = 0
    for i, ch in enumerate(inp):
        if ch not in _CUT5_MARKS:
            continue
        decimal_point = ch == "." and 0 < i < len(inp) - 1 and inp[i - 1].isdigit() and inp[i + 1].isdigit()
        if not decimal_point:
            pieces.append(inp[start:i + 1])
            start = i + 1
    if start < len(inp):
        pieces.append(inp[start:])
    return _keep_spoken(pieces, _CUT5_MARKS)
