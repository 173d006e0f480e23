"""Host-side mirror of the reference's `TextPreprocessor` (GPT_SoVITS/TTS_infer_pack/TextPreprocessor.py:51-239): text ->
list of {phones, bert_features, norm_text} segments for `TTS.run`.

Kept from the reference: consecutive-punctuation collapse (:235-239), the leading-stop rule for very short first
sentences (:82-83), the cut method, blank / symbol-only line filtering, short-segment merging (threshold 5, :32-48),
the trailing stop, the > 510-character split (:93-117), the per-language segment loop with its en / non-en merging rule
(:122-172), the "fewer than 6 phones -> retry with a leading '.'" rule (:187-188), zero BERT features for every language
but zh (:216-220) and the phone-level repetition of zh BERT features by `word2ph` (:199-204).

Plug-ins (the reference hard-wires third-party packages here):
  * `lang_segmenter(text, default_lang) -> [{"lang", "text"}]`  -- reference: LangSegmenter (split_lang + fast_langdetect +
    jieba).  Default: `script_segmenter`, a Unicode-script splitter (Latin -> "en", kana -> "ja", hangul -> "ko", Han ->
    the caller's language).  PARITY UNPINNED against LangSegmenter.
  * G2P per language -- `gsv.text.cleaner.register_g2p`.
  * `bert_fn(norm_text) -> Tensor[len(norm_text), 1024]`        -- reference: chinese-roberta-wwm-ext-large hidden_states[-3]
    with [CLS] / [SEP] dropped (:191-198).  `gsv.feature_extractor.bert` provides the HIP engine for it.
"""
from __future__ import annotations

import re
import threading
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import torch

from ..text import cleaned_text_to_sequence
from ..text.cleaner import clean_text
from .text_segmentation_method import get_method as get_seg_method, split_big_text, splits

punctuation = set(["!", "?", "…", ",", ".", "-"])


def get_first(text: str) -> str:
    pattern = "[" + "".join(re.escape(sep) for sep in splits) + "]"
    return re.split(pattern, text)[0].strip()


def _merged(pieces: Iterable[str], threshold: int) -> Iterator[str]:
    """yield runs of consecutive pieces, each run closed as soon as it holds `threshold` characters; what is left over at
    the end is yielded too (the caller decides where it goes)"""
    run = ""
    for piece in pieces:
        run += piece
        if len(run) >= threshold:
            yield run
            run = ""
    if run:
        yield run


def merge_short_text_in_array(texts: List[str], threshold: int) -> List[str]:
    """reference TextPreprocessor.py:32-48: short pieces are glued to their successors until a run reaches `threshold`
    characters; a short tail joins the last full run (or stands alone when there is none)"""
    if len(texts) < 2:
        return texts
    runs = list(_merged(texts, threshold))
    if len(runs) > 1 and len(runs[-1]) < threshold:
        tail = runs.pop()
        runs[-1] += tail
    return runs


_KANA = re.compile(r"[ぁ-ゖ゙゚ァ-ヺー]")
_HANGUL = re.compile(r"[ᄀ-ᇿ㄰-㆏가-힯]")
_HAN = re.compile(r"[㐀-䶵一-鿿]")
_LATIN = re.compile(r"[A-Za-z]")


def script_segmenter(text: str, default_lang: Optional[str] = None) -> List[Dict[str, str]]:
    """Split at script changes; digits, spaces and punctuation stay with the run they follow."""
    out: List[Dict[str, str]] = []
    cur, buf = None, ""
    for ch in text:
        if _LATIN.match(ch):
            lang = "en"
        elif _KANA.match(ch):
            lang = "ja"
        elif _HANGUL.match(ch):
            lang = "ko"
        elif _HAN.match(ch):
            lang = default_lang if default_lang in ("zh", "ja", "ko", "yue") else "zh"
        else:
            lang = cur
        if lang is None:
            buf += ch
            continue
        if cur is None:
            cur = lang
        if lang != cur:
            out.append({"lang": cur, "text": buf})
            cur, buf = lang, ""
        buf += ch
    if buf:
        out.append({"lang": cur or (default_lang or "en"), "text": buf})
    return out


class TextPreprocessor:
    def __init__(self, bert_fn: Optional[Callable[[str], torch.Tensor]] = None, device="cpu",
                 lang_segmenter: Callable[[str, Optional[str]], List[Dict[str, str]]] = script_segmenter):
        self.bert_fn = bert_fn
        self.device = torch.device(device)
        self.lang_segmenter = lang_segmenter
        self.bert_lock = threading.RLock()

    # ---- reference :59-77
    def preprocess(self, text: str, lang: str, text_split_method: str, version: str = "v2") -> List[Dict]:
        text = self.replace_consecutive_punctuation(text)
        texts = self.pre_seg_text(text, lang, text_split_method)
        result = []
        for text in texts:
            phones, bert_features, norm_text = self.segment_and_extract_feature_for_text(text, lang, version)
            if phones is None or norm_text == "":
                continue
            result.append({"phones": phones, "bert_features": bert_features, "norm_text": norm_text})
        return result

    # ---- reference :79-117, as a chain of small stages
    def pre_seg_text(self, text: str, lang: str, text_split_method: str) -> List[str]:
        stop = "." if lang == "en" else "。"
        text = text.strip("\n")
        if not text:
            return []
        if text[0] not in splits and len(get_first(text)) < 4:      # a very short first sentence gets a leading stop
            text = stop + text
        cut = re.sub(r"\n{2,}", "\n", get_seg_method(text_split_method)(text))
        pieces = merge_short_text_in_array(self.filter_text(cut.split("\n")), 5)

        def speakable(items):
            for t in items:
                if t.strip() and re.sub(r"\W+", "", t):              # not blank, not symbols only
                    yield t

        def terminated(items):
            for t in items:
                yield t if t[-1] in splits else t + stop

        def bounded(items):
            for t in items:
                if len(t) > 510:                                     # BERT's input limit
                    yield from split_big_text(t)
                else:
                    yield t

        return list(bounded(terminated(speakable(pieces))))

    def segment_and_extract_feature_for_text(self, text: str, language: str, version: str = "v1"):
        return self.get_phones_and_bert(text, language, version)

    # ---- reference :122-190
    # How a request language is routed to (language, text) runs, as data: the hint given to the segmenter for Han characters,
    # and what a detected "zh" run is relabelled to.
    _MONO = {"all_zh": ("zh", None), "all_ja": ("ja", None), "all_ko": ("ko", None), "all_yue": ("zh", "yue"),
             "auto": (None, None), "auto_yue": (None, "yue")}

    def _runs(self, text: str, language: str) -> List[Tuple[str, str]]:
        if language == "en":
            return [("en", text)]
        if language in self._MONO:
            hint, zh_as = self._MONO[language]
            return [(zh_as if zh_as and seg["lang"] == "zh" else seg["lang"], seg["text"]) for seg in self.lang_segmenter(text, hint)]
        # a plain language ("zh", "ja", ...): English runs stay English, every other run is the caller's language -- Han characters
        # of zh / ja / ko cannot be told apart -- and neighbours of the same kind are glued together
        runs: List[Tuple[str, str]] = []
        for seg in self.lang_segmenter(text, None):
            kind = "en" if seg["lang"] == "en" else language
            if runs and (runs[-1][0] == "en") == (kind == "en"):
                runs[-1] = (runs[-1][0], runs[-1][1] + seg["text"])
            else:
                runs.append((kind, seg["text"]))
        return runs

    def get_phones_and_bert(self, text: str, language: str, version: str, final: bool = False):
        with self.bert_lock:
            text = re.sub(r" {2,}", " ", text)
            phones: List[int] = []
            berts: List[torch.Tensor] = []
            norm_text = ""
            for lang, piece in self._runs(text, language):
                ph, word2ph, norm = self.clean_text_inf(piece, lang, version)
                berts.append(self.get_bert_inf(ph, word2ph, norm, lang))
                phones += ph
                norm_text += norm
            if not final and len(phones) < 6:                       # too short to synthesise: once more behind a leading stop
                return self.get_phones_and_bert("." + text, language, version, final=True)
            return phones, torch.cat(berts, dim=1), norm_text

    # ---- reference :191-204
    def get_bert_feature(self, text: str, word2ph: list) -> torch.Tensor:
        if self.bert_fn is None:
            raise NotImplementedError("zh text needs a bert_fn (chinese-roberta-wwm-ext-large hidden_states[-3]); "
                                      "see gsv.feature_extractor.bert")
        res = self.bert_fn(text).float().cpu()             # [len(text), 1024]: [CLS] / [SEP] already dropped
        assert len(word2ph) == len(text) == res.shape[0]
        rep = torch.as_tensor(word2ph, dtype=torch.long)
        return torch.repeat_interleave(res, rep, dim=0).T

    def clean_text_inf(self, text: str, language: str, version: str = "v2"):
        language = language.replace("all_", "")
        phones, word2ph, norm_text = clean_text(text, language, version)
        return cleaned_text_to_sequence(phones, version), word2ph, norm_text

    def get_bert_inf(self, phones: list, word2ph: list, norm_text: str, language: str) -> torch.Tensor:
        language = language.replace("all_", "")
        if language == "zh":
            return self.get_bert_feature(norm_text, word2ph).to(self.device)
        return torch.zeros((1024, len(phones)), dtype=torch.float32, device=self.device)

    # ---- reference :222-239
    _EMPTY = (None, " ", "")

    def filter_text(self, texts):
        kept = [t for t in texts if t not in self._EMPTY]
        if not any(t != "\n" for t in kept):
            raise ValueError("请输入有效文本")
        return kept

    _REPEATED_PUNCT = re.compile("([{0}])[{0}]+".format("".join(re.escape(c) for c in sorted(punctuation))))

    def replace_consecutive_punctuation(self, text: str) -> str:
        return self._REPEATED_PUNCT.sub(r"\1", text)
