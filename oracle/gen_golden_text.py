"""TEST INFRASTRUCTURE ONLY -- generates the text front-end fixtures from the reference's own modules (build container only):

  gpt-sovits_amd/gsv/text/symbols_v1.json, symbols_v2.json   the symbol tables as DATA (reference text/symbols.py:399,
                                                             text/symbols2.py:419: `sorted(set(...))` of the phoneme inventories)
  tests/golden/text_segmentation.json                        outputs of TTS_infer_pack/text_segmentation_method.py
  tests/golden/text_preprocess.json                          outputs of the pure-Python parts of TTS_infer_pack/TextPreprocessor.py
                                                             (replace_consecutive_punctuation, pre_seg_text, merge_short_text_in_array,
                                                             filter_text) and of text/__init__.py cleaned_text_to_sequence

The reference's TextPreprocessor module imports LangSegmenter (jieba, fast_langdetect, split_lang), text.chinese (pypinyin, jieba),
text.cleaner and tools.i18n at module scope; none of them is used by the functions pinned here, so they are replaced by empty
stubs.  G2P itself (text/english.py: g2p_en + nltk + wordsegment; text/chinese2.py: pypinyin + g2pw) is NOT importable here:
phoneme strings are inputs of the fixtures, never outputs ("parity unpinned" for G2P).

    python oracle/gen_golden_text.py
"""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/GPT_SoVITS"
sys.path.insert(0, ROOT)

SEG_TEXTS_KEY = "texts"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


PRE_TEXTS = [
    ("Hello there. This is version 2.5 of the system, tested on 3.14 inputs! Is it fine? Yes... it is.", "en"),
    ("Hi", "en"),
    ("a,,.b!!?c", "en"),
    ("No punctuation at all in this sentence", "en"),
    ("\n\nFirst line.\n\n\nSecond line without stop\nThird!\n", "en"),
    ("你好，我是小明。你好，我是小红。你好，我是小刚。你好，我是小张。", "zh"),
    ("一二三四五六七八九十，一二三四五六七八九十。一二三四五六七八九十！一二三四五六七八九十？末尾没有标点", "zh"),
    ("好", "zh"),
    ("......", "en"),
    ("Mixed，标点. with 中文 and English! 还有...省略号…… and a dash - here.", "zh"),
    ("word " * 140, "en"),                                   # > 510 characters in one sentence: split_big_text
    ("短。短。短。短。这是一个稍微长一点的句子，用来凑够长度。短。", "zh"),
]
METHODS = ["cut0", "cut1", "cut2", "cut3", "cut4", "cut5"]


def main():
    # ---- symbol tables as data
    sys.path.insert(0, REF)
    from text import symbols as s1, symbols2 as s2        # plain lists, no third-party imports
    from text import cleaned_text_to_sequence
    out_dir = os.path.join(ROOT, "gpt-sovits_amd", "gsv", "text")
    os.makedirs(out_dir, exist_ok=True)
    for name, mod in (("symbols_v1.json", s1), ("symbols_v2.json", s2)):
        with open(os.path.join(out_dir, name), "w", encoding="utf-8") as f:
            json.dump({"symbols": list(mod.symbols), "punctuation": list(mod.punctuation), "pad": mod.pad}, f, ensure_ascii=False)
    print("symbols:", len(s1.symbols), len(s2.symbols))

    # ---- text_segmentation_method (stand-alone module)
    seg = _load("ref_text_segmentation_method", os.path.join(REF, "TTS_infer_pack", "text_segmentation_method.py"))
    gpath = os.path.join(ROOT, "tests", "golden", "text_segmentation.json")
    old = json.load(open(gpath, encoding="utf-8"))
    texts = old[SEG_TEXTS_KEY]
    new = {"texts": texts,
           "methods": {m: [seg.get_method(m)(t) for t in texts] for m in seg.get_method_names()},
           "split": [seg.split(t) for t in texts],
           "split_big_text": [seg.split_big_text(t, 40) for t in texts]}     # max_len 40 so that the 12 texts exercise the splitting
    assert new == old, "text_segmentation.json differs from what the reference produces now"
    json.dump(new, open(gpath, "w", encoding="utf-8"), ensure_ascii=False, indent=0)

    # ---- TextPreprocessor's pure parts
    class _NoSeg:
        @staticmethod
        def getTexts(text, lang=None):
            raise RuntimeError("LangSegmenter is stubbed")
    pkg = _stub("TTS_infer_pack")
    pkg.__path__ = [os.path.join(REF, "TTS_infer_pack")]
    sys.modules["TTS_infer_pack.text_segmentation_method"] = seg
    _stub("text.LangSegmenter", LangSegmenter=_NoSeg)
    _stub("text.chinese")
    _stub("text.cleaner", clean_text=None)
    _stub("tools")
    _stub("tools.i18n")
    _stub("tools.i18n.i18n", I18nAuto=lambda language=None: (lambda s: s), scan_language_list=lambda: [])
    tp_mod = _load("TTS_infer_pack.TextPreprocessor", os.path.join(REF, "TTS_infer_pack", "TextPreprocessor.py"))
    tp = tp_mod.TextPreprocessor(None, None, "cpu")
    import contextlib
    import io
    cases = []
    for text, lang in PRE_TEXTS:
        for m in METHODS:
            with contextlib.redirect_stdout(io.StringIO()):
                rep = tp.replace_consecutive_punctuation(text)
                try:
                    segs = tp.pre_seg_text(rep, lang, m)
                    err = None
                except ValueError as e:
                    segs, err = None, "ValueError"
            cases.append({"text": text, "lang": lang, "method": m, "replaced": rep, "segments": segs, "error": err})
    merge = [(xs, th, tp_mod.merge_short_text_in_array(list(xs), th))
             for xs, th in [(["a", "bb", "ccc", "dddddd"], 5), (["abc"], 5), ([], 5), (["ab", "cd"], 5), (["abcdef", "g"], 5),
                            (["一", "二三四五六", "七"], 5)]]
    first = [(t, tp_mod.get_first(t)) for t in ["Hi, there", "你好。再见", "nopunct", ".lead"]]
    seqs = []
    for ver, mod in (("v1", s1), ("v2", s2)):
        ph = [mod.symbols[i] for i in range(0, len(mod.symbols), 7)] + ["UNK", ",", ".", "SP"]
        seqs.append({"version": ver, "phones": ph, "ids": cleaned_text_to_sequence(ph, ver)})
    out = {"pre_seg": cases, "merge_short": merge, "get_first": first, "cleaned_text_to_sequence": seqs}
    with open(os.path.join(ROOT, "tests", "golden", "text_preprocess.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("pre_seg cases:", len(cases))


if __name__ == "__main__":
    main()
