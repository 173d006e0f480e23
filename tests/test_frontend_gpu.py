"""GPU: the reference-audio front-end through the C ABI's op entry points:
`spectrogram_torch` against the reference function's outputs (tests/golden/spec_*.npz), the HuBERT-base engine against
`transformers.HubertModel` (the class the reference's CNHubert wraps) on synthetic weights (tests/golden/hubert_base.npz),
and `TTS.set_ref_audio` end to end from a WAV file."""
import os
import wave

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name,n,seed", [("spec_32k_half_s", 16000, 0), ("spec_32k_ragged", 22001, 1), ("spec_short", 1500, 2)])
def test_spectrogram_matches_reference(name, n, seed):
    """fp32 DFT-as-GEMM vs torch.stft in the reference function: values up to ~75; max-abs <= 5e-4, relative rms <= 5e-6."""
    from gsv import synthetic as S
    from gsv.module.mel_processing import spectrogram_torch
    y = S.make_waveform(n, seed, sr=32000).unsqueeze(0).to(DEV)
    spec = spectrogram_torch(y, 2048, 32000, 640, 2048, center=False).cpu().numpy()
    g = load_golden(name)["spec"]
    assert spec.shape == g.shape
    err = np.abs(spec - g).max()
    rel = np.sqrt(((spec - g) ** 2).mean() / (g ** 2).mean())
    print(f"[frontend] {name}: max-abs {err:.2e}, relative rms {rel:.2e}")
    assert err <= 5e-4 and rel <= 5e-6
    with pytest.raises(ValueError):
        spectrogram_torch(y[:, :500], 2048, 32000, 640, 2048)          # shorter than the reflect padding
    with pytest.raises(NotImplementedError):
        spectrogram_torch(y, 2048, 32000, 640, 2048, center=True)


@pytest.mark.parametrize("name,n_fft,hop,sr,n,seed", [("mel_v3", 1024, 256, 24000, 36000, 3), ("mel_v4", 1280, 320, 32000, 41003, 4)])
def test_mel_spectrogram_matches_reference(name, n_fft, hop, sr, n, seed):
    """v3 / v4 `mel_fn` (TTS.py:67-93): DFT GEMM + frame-major magnitude + filterbank GEMM with the log-clamp epilogue vs the
    reference's mel_spectrogram_torch (fp32 torch.stft); log-mel values in [-6.7, 0.5]: max-abs <= 2e-3, rms <= 1e-4."""
    from gsv import synthetic as S
    from gsv.TTS_infer_pack import TTS as T
    y = S.make_waveform(n, seed, sr=sr).unsqueeze(0).to(DEV)
    mel = (T.mel_fn if name == "mel_v3" else T.mel_fn_v4)(y)
    g = load_golden(name)["mel"]
    o = mel.cpu().numpy()
    assert o.shape == g.shape and mel.dtype == torch.float32
    err, rms = np.abs(o - g).max(), np.sqrt(((o - g) ** 2).mean())
    print(f"[frontend] {name}: max-abs {err:.2e}, rms {rms:.2e}")
    assert err <= 2e-3 and rms <= 1e-4


def test_hubert_matches_transformers_model():
    """fp16 engine vs transformers.HubertModel fp32 on the same synthetic weights and waveform: last_hidden_state of rms 1.0,
    64 frames x 768; bar: relative rms <= 0.5 %, max-abs <= 3e-2 (measured 0.14 %, 6.8e-3) (fp16 activations through 7 convs + 12 post-LN layers)."""
    from gsv import synthetic as S
    from gsv.feature_extractor.cnhubert import CNHubert
    g = load_golden("hubert_base")
    ref = g["last_hidden_state"].astype(np.float32)
    m = CNHubert(device=DEV, state_dict=S.make_hubert_state_dict(seed=0))
    wav = S.make_waveform(20800, 7).unsqueeze(0).to(DEV)
    out = m.model(wav)["last_hidden_state"]
    assert tuple(out.shape) == (1, 64, 768)
    o = out[0].float().cpu().numpy()
    rel = np.sqrt(((o - ref) ** 2).mean() / (ref ** 2).mean())
    err = np.abs(o - ref).max()
    print(f"[frontend] HuBERT last_hidden_state: relative rms {rel:.3e}, max-abs {err:.3e}")
    assert rel <= 5e-3 and err <= 3e-2
    out2 = m.model(wav)["last_hidden_state"]
    assert torch.equal(out, out2)
    with pytest.raises(ValueError):
        m.model(wav[:, :300])


def _write_wav(path, x, sr):
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def test_set_ref_audio_end_to_end(tmp_path):
    """WAV file -> resample -> HuBERT -> ssl_proj + VQ codes (prompt_semantic) and spectrogram (refer_spec), then a whole
    TTS.run on that prompt; the 3-10 s guard raises OSError like the reference (TTS.py:802-803)."""
    from gsv import synthetic as S
    from gsv.TTS_infer_pack.TTS import TTS
    tts = TTS({"device": DEV, "is_half": True, "version": "v2", "max_batch": 4, "max_seq": 600})
    tts.init_t2s_weights(state={"weight": S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0, suppress_eos=True), "config": S.T2S_V2_CONFIG})
    tts.init_vits_weights(state={"weight": S.make_vits_state_dict(S.VITS_V2_CONFIG, seed=0), "config": dict(S.VITS_V2_CONFIG)})
    tts.init_cnhuhbert_weights(state_dict=S.make_hubert_state_dict(seed=0))
    wav = S.make_waveform(4 * 24000, 3, sr=24000).numpy()
    p = str(tmp_path / "ref.wav")
    _write_wav(p, wav, 24000)
    tts.set_ref_audio(p)
    ps = tts.prompt_cache["prompt_semantic"]
    # 4 s + 0.3 s * 32000 / 16000 silence samples = 73600 samples at 16 kHz -> 229 HuBERT frames -> 114 codes
    assert ps.dtype == torch.int64 and ps.dim() == 1 and ps.numel() == 114 and int(ps.max()) < 1024
    spec = tts.prompt_cache["refer_spec"][0][0]
    assert tuple(spec.shape) == (1, 1025, 200) and spec.dtype == torch.float16
    short = str(tmp_path / "short.wav")
    _write_wav(short, wav[: 2 * 24000], 24000)
    with pytest.raises(OSError):
        tts.set_ref_audio(short)
    utt = S.make_utterances(2)
    tts.prompt_cache["phones"] = utt["prompt_phones"]
    tts.prompt_cache["bert_features"] = torch.zeros(1024, len(utt["prompt_phones"]))
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": it["norm_text"]}
            for it in utt["items"]]
    tts.configs.max_sec = 0.4
    out = list(tts.run({"segments": segs, "batch_size": 2, "top_k": 1, "seed": 0}))
    sr, audio = out[-1]
    assert sr == 32000 and audio.dtype == np.int16 and audio.size == 2 * (20 * 1280 + int(32000 * 0.3))
    # auxiliary references (TTS.py:1098-1113): spectrograms appended after the main one, cached by path set, missing files skipped
    aux = str(tmp_path / "aux.wav")
    _write_wav(aux, S.make_waveform(2 * 32000, 9, sr=32000).numpy(), 32000)
    req = {"segments": segs, "batch_size": 2, "top_k": 1, "seed": 0, "aux_ref_audio_paths": [aux, str(tmp_path / "missing.wav")]}
    sr, audio2 = list(tts.run(req))[-1]
    assert len(tts.prompt_cache["refer_spec"]) == 2 and tuple(tts.prompt_cache["refer_spec"][1][0].shape) == (1, 1025, 100)
    assert audio2.shape == audio.shape and not np.array_equal(audio2, audio)          # the style vector changed
    kept = tts.prompt_cache["refer_spec"][1][0]
    list(tts.run(req))
    assert tts.prompt_cache["refer_spec"][1][0] is kept                              # same path set: nothing recomputed
    sr, audio3 = list(tts.run(dict(req, aux_ref_audio_paths=[])))[-1]
    assert len(tts.prompt_cache["refer_spec"]) == 1 and np.array_equal(audio3, audio)


def test_bert_feature_matches_transformers_model():
    """zh BERT features: fp16 engine (22 of 24 BERT-large layers) vs transformers.BertModel fp32 hidden_states[-3][1:-1] on the
    same synthetic weights and a 26-character string (rms 1.0): relative rms <= 1 %, max-abs <= 5e-2; one row per character,
    unknown characters map to [UNK], more than 510 characters are refused."""
    from gsv import synthetic as S
    from gsv.feature_extractor.bert import BertFeature
    text = "你好，我是小明。今天天气不错，我们一起去公园散步吧！"
    ref = load_golden("bert_large")["feature"].astype(np.float32)
    bf = BertFeature(device=DEV, state_dict=S.make_bert_state_dict(seed=0), vocab=S.BERT_TEST_VOCAB)
    out = bf(text)
    assert tuple(out.shape) == (len(text), 1024) == ref.shape
    o = out.cpu().numpy()
    rel = np.sqrt(((o - ref) ** 2).mean() / (ref ** 2).mean())
    err = np.abs(o - ref).max()
    print(f"[frontend] BERT hidden_states[-3]: relative rms {rel:.3e}, max-abs {err:.3e}")
    assert rel <= 1e-2 and err <= 5e-2
    assert bf.tokenize("你?") == [2, S.BERT_TEST_VOCAB.index("你"), 1, 3]          # '?' (ASCII) is not in the test vocabulary
    with pytest.raises(ValueError):
        bf("好" * 511)


def test_zh_text_through_preprocessor_with_bert_engine():
    """the plug-in seam end to end: TextPreprocessor (zh) -> BertFeature -> phone-level features [1024, n_phones]"""
    from gsv import synthetic as S
    from gsv.feature_extractor.bert import BertFeature
    from gsv.text import cleaner
    from gsv.TTS_infer_pack.TextPreprocessor import TextPreprocessor

    class Zh:
        def text_normalize(self, t):
            return t

        def g2p(self, norm):
            ph, w2p = [], []
            for ch in norm:
                if ch in "，。！？":
                    ph.append({"，": ",", "。": ".", "！": "!", "？": "?"}[ch]); w2p.append(1)
                else:
                    ph += ["n", "i3"]; w2p.append(2)
            return ph, w2p
    cleaner.register_g2p("zh", Zh())
    bf = BertFeature(device=DEV, state_dict=S.make_bert_state_dict(seed=0, layers=22), vocab=S.BERT_TEST_VOCAB)
    tp = TextPreprocessor(bert_fn=bf)
    segs = tp.preprocess("你好，我是小明。今天天气不错！", "all_zh", "cut5", "v2")
    assert [sg["norm_text"] for sg in segs] == ["你好，我是小明。", "今天天气不错！"]       # "你好，" (< 5 characters) merges forward
    for sg in segs:
        assert sg["bert_features"].shape == (1024, len(sg["phones"])) and bool(sg["bert_features"].any())
    f = segs[0]["bert_features"]
    assert torch.equal(f[:, 0], f[:, 1]) and not torch.equal(f[:, 1], f[:, 2])      # two phones of one character share its vector


def test_kaldi_fbank_matches_reference():
    """N4: Kaldi fbank with the frame pre-processing folded into the DFT basis vs the reference's eres2net/kaldi.py (fp32 torch):
    298 frames x 80 log-energies in [-13.4, 1.2]; max-abs <= 2e-3, rms <= 1e-4; short input -> empty result (kaldi.py:65)."""
    from gsv import synthetic as S
    from gsv.eres2net import kaldi as Kaldi
    g = load_golden("sv_eres2net")["fbank"]
    wav = S.make_waveform(48000, 5).unsqueeze(0).to(DEV)
    fb = Kaldi.fbank(wav, num_mel_bins=80, sample_frequency=16000, dither=0)
    o = fb.cpu().numpy()
    assert o.shape == g.shape == (298, 80)
    err, rms = np.abs(o - g).max(), np.sqrt(((o - g) ** 2).mean())
    print(f"[frontend] kaldi fbank: max-abs {err:.2e}, rms {rms:.2e}")
    assert err <= 2e-3 and rms <= 1e-4
    assert Kaldi.fbank(wav[:, :300], num_mel_bins=80).shape == (0, 80)
    with pytest.raises(NotImplementedError):
        Kaldi.fbank(wav, num_mel_bins=80, dither=1.0)


def test_eres2netv2_embedding_matches_reference():
    """N4: SV.compute_embedding3 (fbank -> ERes2NetV2 w24 s4 e4 forward3, fp32 engine, BatchNorm folded) vs the reference classes
    on the same synthetic weights: 20480 values of mean |.| 2.0; relative rms <= 1e-4, max-abs <= 2e-3.  Also block by block
    against the oracle's taps on a shorter input."""
    from gsv import synthetic as S
    from gsv.sv import SV
    from oracle import sv_oracle as so
    g = load_golden("sv_eres2net")["emb"]
    sd = S.make_eres2net_state_dict(seed=0)
    sv = SV(DEV, False, state_dict=sd)
    wav = S.make_waveform(48000, 5).unsqueeze(0).to(DEV)
    emb = sv.compute_embedding3(wav)
    assert tuple(emb.shape) == (1, 20480) and emb.dtype == torch.float32
    o = emb[0].cpu().numpy()
    rel, err = np.sqrt(((o - g) ** 2).mean() / (g ** 2).mean()), np.abs(o - g).max()
    print(f"[frontend] ERes2NetV2 embedding: relative rms {rel:.2e}, max-abs {err:.2e}")
    assert rel <= 1e-4 and err <= 2e-3
    assert SV(DEV, True, state_dict=sd).compute_embedding3(wav.half()).dtype == torch.float16
    # odd frame count (stride-2 stages round up) and a batch of two
    wav2 = torch.stack([S.make_waveform(16000 + 570, 6), S.make_waveform(16000 + 570, 7)])
    ref = so.compute_embedding3(sd, wav2)
    out = sv.compute_embedding3(wav2.to(DEV)).cpu()
    assert out.shape == ref.shape and ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= 1e-4


def test_set_ref_audio_v2pro_computes_speaker_embedding(tmp_path):
    """v2Pro: set_ref_audio also produces the ERes2NetV2 embedding of the (peak-normalised, 16 kHz) reference audio
    (TTS.py:790-800, 1233-1238) and TTS.run synthesises with it; without init_sv_model() the call is refused."""
    import copy
    from gsv import synthetic as S
    from gsv.TTS_infer_pack.TTS import TTS
    from oracle import sv_oracle as so
    cfg = copy.deepcopy(S.VITS_V2_CONFIG)
    cfg["model"]["version"], cfg["model"]["gin_channels"] = "v2Pro", 1024
    tts = TTS({"device": DEV, "is_half": True, "version": "v2Pro", "max_batch": 4, "max_seq": 600})
    tts.init_t2s_weights(state={"weight": S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0, suppress_eos=True), "config": S.T2S_V2_CONFIG})
    tts.init_vits_weights(state={"weight": S.make_vits_state_dict(cfg, seed=0), "config": cfg})
    tts.init_cnhuhbert_weights(state_dict=S.make_hubert_state_dict(seed=0))
    wav = S.make_waveform(3 * 32000 + 777, 3, sr=32000).numpy()
    p = str(tmp_path / "ref.wav")
    _write_wav(p, wav, 32000)
    with pytest.raises(RuntimeError):
        tts.set_ref_audio(p)                                         # no SV model yet
    sd = S.make_eres2net_state_dict(seed=0)
    tts.init_sv_model(state_dict=sd)
    tts.set_ref_audio(p)
    spec, audio16k = tts.prompt_cache["refer_spec"][0]
    emb = tts.prompt_cache["sv_emb"][0]
    assert audio16k.dtype == torch.float16 and tuple(emb.shape) == (1, 20480) and emb.dtype == torch.float16
    ref = so.compute_embedding3(sd, audio16k.float().cpu())
    rel = float(((emb.float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    assert rel <= 2e-3                                               # the fp16 cast of the result dominates
    utt = S.make_utterances(2)
    tts.prompt_cache["phones"] = utt["prompt_phones"]
    tts.prompt_cache["bert_features"] = torch.zeros(1024, len(utt["prompt_phones"]))
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": it["norm_text"]}
            for it in utt["items"]]
    tts.configs.max_sec = 0.4
    sr, audio = list(tts.run({"segments": segs, "batch_size": 2, "top_k": 1, "seed": 0}))[-1]
    assert sr == 32000 and audio.dtype == np.int16 and audio.size == 2 * (20 * 1280 + int(32000 * 0.3)) and np.abs(audio).max() > 0
