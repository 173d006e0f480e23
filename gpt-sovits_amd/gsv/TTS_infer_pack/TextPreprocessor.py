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
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ..text import cleaned_text_to_sequence
from ..text.cleaner import clean_text
from .text_segmentation_method import get_method as get_seg_method, split_big_text, splits

punctuation = set(["!", "?", "…", ",", ".", "-"])


def get_first(text: str) -> str:
    pattern = "[" + "".join(re.escape(sep) for sep in splits) + "]"
    return re.split(pattern, text)[0].strip()


def merge_short_text_in_array(texts: List[str], threshold: int) -> List[str]:
    if len(texts) < 2:
        return texts
    result, text = [], ""
    for ele in texts:
        text += ele
        if len(text) >= threshold:
            result.append(text)
            text = ""
    if len(text) > 0:
        if len(result) == 0:
            result.append(text)
        else:
            result[-1] += text
    return result


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

    # ---- reference :79-117
    def pre_seg_text(self, text: str, lang: str, text_split_method: str) -> List[str]:
        text = text.strip("\n")
        if len(text) == 0:
            return []
        if text[0] not in splits and len(get_first(text)) < 4:
            text = "。" + text if lang != "en" else "." + text
        text = get_seg_method(text_split_method)(text)
        while "\n\n" in text:
            text = text.replace("\n\n", "\n")
        _texts = merge_short_text_in_array(self.filter_text(text.split("\n")), 5)
        texts: List[str] = []
        for text in _texts:
            if len(text.strip()) == 0:
                continue
            if not re.sub(r"\W+", "", text):
                continue                                   # symbols only
            if text[-1] not in splits:
                text += "。" if lang != "en" else "."
            if len(text) > 510:                            # BERT's input limit
                texts.extend(split_big_text(text))
            else:
                texts.append(text)
        return texts

    def segment_and_extract_feature_for_text(self, text: str, language: str, version: str = "v1"):
        return self.get_phones_and_bert(text, language, version)

    # ---- reference :122-190
    def get_phones_and_bert(self, text: str, language: str, version: str, final: bool = False):
        with self.bert_lock:
            text = re.sub(r" {2,}", " ", text)
            textlist: List[str] = []
            langlist: List[str] = []
            seg = self.lang_segmenter
            if language in ("all_zh", "all_yue", "all_ja", "all_ko"):
                base = language[4:]
                for tmp in seg(text, "zh" if base == "yue" else base):
                    lang = tmp["lang"]
                    if base == "yue" and lang == "zh":
                        lang = "yue"
                    langlist.append(lang)
                    textlist.append(tmp["text"])
            elif language == "en":
                langlist.append("en")
                textlist.append(text)
            elif language in ("auto", "auto_yue"):
                for tmp in seg(text, None):
                    lang = tmp["lang"]
                    if language == "auto_yue" and lang == "zh":
                        lang = "yue"
                    langlist.append(lang)
                    textlist.append(tmp["text"])
            else:
                for tmp in seg(text, None):
                    if langlist:
                        if (tmp["lang"] == "en" and langlist[-1] == "en") or (tmp["lang"] != "en" and langlist[-1] != "en"):
                            textlist[-1] += tmp["text"]
                            continue
                    # Han characters of zh / ja / ko cannot be told apart: the caller's language decides
                    langlist.append("en" if tmp["lang"] == "en" else language)
                    textlist.append(tmp["text"])
            phones_list, bert_list, norm_text_list = [], [], []
            for i in range(len(textlist)):
                lang = langlist[i]
                phones, word2ph, norm_text = self.clean_text_inf(textlist[i], lang, version)
                bert_list.append(self.get_bert_inf(phones, word2ph, norm_text, lang))
                phones_list.append(phones)
                norm_text_list.append(norm_text)
            bert = torch.cat(bert_list, dim=1)
            phones = sum(phones_list, [])
            norm_text = "".join(norm_text_list)
            if not final and len(phones) < 6:
                return self.get_phones_and_bert("." + text, language, version, final=True)
            return phones, bert, norm_text

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
    def filter_text(self, texts):
        if all(text in [None, " ", "\n", ""] for text in texts):
            raise ValueError("请输入有效文本")
        return [t for t in texts if t not in [None, " ", ""]]

    def replace_consecutive_punctuation(self, text: str) -> str:
        p = "".join(re.escape(c) for c in punctuation)
        return re.sub(f"([{p}])([{p}])+", r"\1", text)
