"""CPU: the text front-end shim against outputs of the reference's own modules
(tests/golden/text_preprocess.json, written by oracle/gen_golden_text.py from TTS_infer_pack/TextPreprocessor.py and
text/__init__.py) and hand-computed cases for the parts whose reference module cannot be imported (text/cleaner.py needs
the G2P packages: its control flow is restated from cleaner.py:21-83 and exercised with stub back-ends)."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "text_preprocess.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD, encoding="utf-8"))


def test_symbol_tables_and_id_mapping_match_reference(gold):
    from gsv.text import cleaned_text_to_sequence, symbols
    assert len(symbols("v1")) == 322 and len(symbols("v2")) == 732
    assert symbols("v3") is symbols("v2")                                 # anything but v1 uses symbols2
    for c in gold["cleaned_text_to_sequence"]:
        assert cleaned_text_to_sequence(c["phones"], c["version"]) == c["ids"]
    with pytest.raises(KeyError):
        cleaned_text_to_sequence(["not-a-symbol"], "v2")


def test_pre_seg_text_and_helpers_match_reference(gold):
    from gsv.TTS_infer_pack.TextPreprocessor import TextPreprocessor, get_first, merge_short_text_in_array
    tp = TextPreprocessor()
    for c in gold["pre_seg"]:
        rep = tp.replace_consecutive_punctuation(c["text"])
        assert rep == c["replaced"]
        if c["error"]:
            with pytest.raises(ValueError):
                tp.pre_seg_text(rep, c["lang"], c["method"])
        else:
            assert tp.pre_seg_text(rep, c["lang"], c["method"]) == c["segments"], (c["text"][:30], c["method"])
    for xs, th, want in gold["merge_short"]:
        assert merge_short_text_in_array(list(xs), th) == want
    for t, want in gold["get_first"]:
        assert get_first(t) == want


class _StubEn:
    def text_normalize(self, text):
        return text.upper()

    def g2p(self, norm):
        return [w for w in norm.split()]


class _StubZh:
    def text_normalize(self, text):
        return text

    def g2p(self, norm):
        ph, w2p = [], []
        for ch in norm:
            if ch == ",":
                ph.append(","); w2p.append(1)
            else:
                ph += ["n", "i3"]; w2p.append(2)
        return ph, w2p


def test_clean_text_control_flow():
    """cleaner.py:21-83: unknown language -> ("en", " "); fewer than 4 English phones get a leading comma; symbols outside the
    table become UNK; zh returns word2ph with the two length invariants; the SP2 / SP3 marks."""
    from gsv.text import cleaner
    cleaner.register_g2p("en", _StubEn())
    cleaner.register_g2p("zh", _StubZh())
    ph, w2p, norm = cleaner.clean_text("hh ah0 l ow1 zz9", "en", "v2")
    assert ph == ["HH", "AH0", "L", "OW1", "UNK"] and w2p is None and norm == "HH AH0 L OW1 ZZ9"
    ph, _, _ = cleaner.clean_text("hh ay1", "en", "v2")
    assert ph == [",", "HH", "AY1"]
    ph, _, norm = cleaner.clean_text("whatever", "xx", "v2")          # language not in the v2 map
    assert norm == " " and ph == [","]
    ph, w2p, norm = cleaner.clean_text("你好", "zh", "v2")
    assert ph == ["n", "i3", "n", "i3"] and w2p == [2, 2] and len(norm) == len(w2p)
    ph, w2p, _ = cleaner.clean_text("你￥好", "zh", "v2")
    assert ph == ["n", "i3", "SP2", "n", "i3"] and w2p == [2, 1, 2]
    ph, _, _ = cleaner.clean_text("你^好", "zh", "v2")
    assert "SP3" in ph
    ph, _, _ = cleaner.clean_text("한국어", "ko", "v1")                 # ko is not a v1 language -> en, " "
    assert ph == [","]
    with pytest.raises(NotImplementedError):
        cleaner._registry.pop("zh")
        cleaner.clean_text("你好", "zh", "v2")


def test_preprocess_end_to_end_with_builtin_backends(tmp_path):
    from gsv.text import cleaner, g2p
    from gsv.TTS_infer_pack.TextPreprocessor import TextPreprocessor, script_segmenter
    cleaner.register_g2p("en", g2p.SymbolG2P())
    tp = TextPreprocessor()
    segs = tp.preprocess("HH AH0 L OW1 W ER1 L D . DH IH1 S IH1 Z AH0 T EH1 S T !", "en", "cut4", "v2")
    assert [s["norm_text"] for s in segs] == ["HH AH0 L OW1 W ER1 L D .", "DH IH1 S IH1 Z AH0 T EH1 S T !"]
    for sg in segs:
        assert sg["bert_features"].shape == (1024, len(sg["phones"])) and not bool(sg["bert_features"].any())
        assert all(isinstance(i, int) and 0 <= i < 732 for i in sg["phones"])
    # fewer than 6 phones: retried once with a leading "." (TextPreprocessor.py:187-188)
    ph, bert, norm = tp.get_phones_and_bert("HH AY1 .", "en", "v2")
    assert norm.startswith(".") and len(ph) == 4
    # dictionary back-end on a CMUdict-format file
    d = tmp_path / "mini.dict"
    d.write_text(";;; comment\nHELLO  HH AH0 L OW1\nWORLD  W ER1 L D\nA  AH0\nA(1)  EY1\nB  B IY1\nTEST  T EH1 S T\nTWO  T UW1\n"
                 "CAT  K AE1 T\nBUS  B AH1 S\nDOG  D AO1 G\n")
    en = g2p.DictG2P(str(d))
    assert en.g2p(en.text_normalize("Hello, world!")) == ["HH", "AH0", "L", "OW1", ",", "W", "ER1", "L", "D", "!"]
    assert en.g2p("cat's bus's dog's") == ["K", "AE1", "T", "S", "B", "AH1", "S", "AH0", "Z", "D", "AO1", "G", "Z"]
    assert en.g2p("A b") == ["EY1", "B", "IY1"]                       # single letters; capital A is EY1 (english.py:285-289)
    assert en.text_normalize("test 2；ok") == "test two ,ok"
    # script segmenter: Latin runs are en, Han runs take the caller's language, kana is ja
    assert [(x["lang"], x["text"]) for x in script_segmenter("你好hello世界。", "zh")] == [("zh", "你好"), ("en", "hello"), ("zh", "世界。")]
    assert script_segmenter("こんにちはworld", "ja")[0]["lang"] == "ja"
    # zh needs a BERT back-end: features are repeated per phone by word2ph (TextPreprocessor.py:199-204)
    cleaner.register_g2p("zh", _StubZh())
    calls = []

    def bert_fn(text):
        calls.append(text)
        return torch.arange(len(text), dtype=torch.float32).unsqueeze(1).expand(-1, 1024).contiguous()
    tz = TextPreprocessor(bert_fn=bert_fn)
    ph, bert, norm = tz.get_phones_and_bert("你好,世界。", "all_zh", "v2")
    assert calls and bert.shape == (1024, len(ph))
    assert bert[0].tolist()[:5] == [0.0, 0.0, 1.0, 1.0, 2.0]
    with pytest.raises(NotImplementedError):
        TextPreprocessor().get_phones_and_bert("你好,世界。", "all_zh", "v2")


def test_mel_filterbank_known_answers_and_oracle_agreement():
    """`librosa.filters.mel` restated twice (product: vectorised, oracle: scalar loops; the package is not installed -- the
    matrix is "parity unpinned" against it).  Known answers from librosa's documentation: hz_to_mel(60) = 0.9,
    mel_to_hz([1, 2, 3]) = [66.667, 133.333, 200.], mel_frequencies(n_mels=40) (fmax 11025) starts 0, 85.317, 170.635, ... and
    passes 1024.856, 1119.114, 1222.042 after the 1 kHz knee; filters.mel(sr=22050, n_fft=2048)[0, 1] prints as 0.016."""
    import numpy as np
    from gsv.module import mel_processing as mp
    from oracle import mel_filterbank as omf
    assert abs(float(mp._hz_to_mel(60)) - 0.9) < 1e-12 and abs(omf.hz_to_mel(60) - 0.9) < 1e-12
    assert np.allclose(mp._mel_to_hz([1, 2, 3]), [200 / 3, 400 / 3, 200.0])
    mf = omf.mel_frequencies(40, 0.0, 11025.0)
    assert [round(v, 3) for v in mf[:3]] == [0.0, 85.317, 170.635] and [round(v, 3) for v in mf[12:15]] == [1024.856, 1119.114, 1222.042]
    assert mf[-1] == pytest.approx(11025.0)
    m = mp.librosa_mel_fn(22050, 2048)
    assert m.shape == (128, 1025) and m.dtype == np.float32 and round(float(m[0, 1]), 3) == 0.016 and float(m[0, 0]) == 0.0
    for sr, n_fft, n_mels in ((22050, 2048, 128), (24000, 1024, 100), (32000, 1280, 100)):
        a, b = mp.librosa_mel_fn(sr, n_fft, n_mels, 0, None), omf.mel(sr, n_fft, n_mels, 0, None)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 and (a >= 0).all()
        assert (a.sum(1) > 0).all()                                   # no empty band at these sizes


def test_mel_oracle_reproduces_reference_goldens():
    """oracle/mel_filterbank.mel_spectrogram against tests/golden/mel_v3.npz / mel_v4.npz (the reference's own
    mel_spectrogram_torch run in the build container with the oracle filterbank in place of librosa's)"""
    import os
    import numpy as np
    from gsv import synthetic as S
    from oracle import mel_filterbank as omf
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name, (n_fft, hop, sr, n, seed) in {"mel_v3": (1024, 256, 24000, 36000, 3), "mel_v4": (1280, 320, 32000, 41003, 4)}.items():
        g = np.load(os.path.join(gold, name + ".npz"))["mel"]
        y = S.make_waveform(n, seed, sr=sr).unsqueeze(0)
        o = omf.mel_spectrogram(y, n_fft, 100, sr, hop, n_fft, 0, None).numpy()
        assert o.shape == g.shape and np.abs(o - g).max() <= 1e-5


def test_sv_oracle_reproduces_reference_golden():
    """oracle/sv_oracle.py (Kaldi fbank + ERes2NetV2.forward3 restated) against tests/golden/sv_eres2net.npz (the reference's own
    eres2net/kaldi.py and ERes2NetV2 classes run in the build container on the synthetic weights)"""
    import os
    import numpy as np
    import torch
    from gsv import synthetic as S
    from oracle import sv_oracle as so
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sv_eres2net.npz"))
    fb = so.fbank(S.make_waveform(48000, 5))
    assert fb.shape == (298, 80) and np.abs(fb.numpy() - g["fbank"]).max() <= 5e-4
    emb = so.ERes2NetV2Oracle(S.make_eres2net_state_dict(seed=0)).forward3(torch.from_numpy(g["fbank"]).unsqueeze(0))
    assert emb.shape == (1, 20480) and np.abs(emb[0].numpy() - g["emb"]).max() <= 1e-4
