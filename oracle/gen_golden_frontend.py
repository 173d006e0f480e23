"""TEST INFRASTRUCTURE ONLY -- fixtures for the reference-audio front-end, build container only.

  tests/golden/spec_*.npz     the reference's own `spectrogram_torch` (GPT_SoVITS/module/mel_processing.py:40-74, pure torch: it
                              imports here with oracle/ref_import's librosa stub) on seeded waveforms
  tests/golden/bert_large.npz `transformers.BertModel` (BERT-large shape = chinese-roberta-wwm-ext-large's encoder) on
                              gsv.synthetic.make_bert_state_dict weights: hidden_states[-3][1:-1] of a 25-character string
  tests/golden/hubert_base.npz `transformers.HubertModel(HubertConfig())` -- the class the reference's CNHubert wraps
                              (feature_extractor/cnhubert.py:22-37) -- with gsv.synthetic.make_hubert_state_dict weights on a
                              seeded 1.3 s waveform: last_hidden_state (fp16) and a few intermediate checksums.
                              The container has transformers 5.x, the reference pins >= 4.43, <= 4.50 (requirements.txt:21):
                              same architecture and state-dict schema; recorded here because the oracle is that package.

  tests/golden/mel_v3.npz / mel_v4.npz  the reference's `mel_spectrogram_torch` (module/mel_processing.py:93-143) with the v3 /
                              v4 `mel_fn` parameters (TTS_infer_pack/TTS.py:67-92) on seeded waveforms; librosa is not
                              installed, so the module's `librosa_mel_fn` is oracle/mel_filterbank.mel here (the filterbank
                              matrix itself is "parity unpinned", the STFT / magnitude / matmul / log-clamp around it is pinned)
  tests/golden/sv_eres2net.npz  the reference's Kaldi fbank (eres2net/kaldi.py, pure torch) and ERes2NetV2(baseWidth=24, scale=4,
                              expansion=4).forward3 (eres2net/ERes2NetV2.py) -- what SV.compute_embedding3 (sv.py:24-32) runs --
                              on gsv.synthetic.make_eres2net_state_dict weights and a seeded 3 s waveform at 16 kHz

    python oracle/gen_golden_frontend.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
from gsv import synthetic as S  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SPEC_CASES = {"spec_32k_half_s": (16000, 0), "spec_32k_ragged": (22001, 1), "spec_short": (1500, 2)}
BERT_TEXT = "你好，我是小明。今天天气不错，我们一起去公园散步吧！"
SV_N = 48000             # 3 s at 16 kHz -> 298 fbank frames
HUBERT_N = 20800          # 1.3 s at 16 kHz -> 64 frames


def main():
    from transformers import HubertConfig, HubertModel
    torch.manual_seed(0)
    m = HubertModel(HubertConfig()).eval()
    sd = S.make_hubert_state_dict(seed=0)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    assert all("masked_spec_embed" in k for k in missing.missing_keys), missing.missing_keys      # training-only parameter
    wav = S.make_waveform(HUBERT_N, 7).unsqueeze(0)
    with torch.no_grad():
        out = m(wav, output_hidden_states=True)
        feats = m.feature_extractor(wav)                 # [1, 512, T]
    h = out["last_hidden_state"][0]
    np.savez_compressed(os.path.join(GOLD, "hubert_base.npz"), last_hidden_state=h.numpy().astype(np.float16),
                        conv_features_rms=np.float32(feats.pow(2).mean().sqrt().item()),
                        layer_rms=np.array([x.pow(2).mean().sqrt().item() for x in out["hidden_states"]], dtype=np.float32))
    print("hubert", tuple(h.shape), "rms", float(h.pow(2).mean().sqrt()))
    # ---- BERT-large: hidden_states[-3][1:-1] exactly as TextPreprocessor.get_bert_feature takes it (:191-198)
    from transformers import BertConfig, BertModel
    bcfg = BertConfig(vocab_size=len(S.BERT_TEST_VOCAB), hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                      intermediate_size=4096, max_position_embeddings=512)
    bm = BertModel(bcfg, add_pooling_layer=False).eval()
    miss = bm.load_state_dict(S.make_bert_state_dict(seed=0), strict=False)
    assert not miss.unexpected_keys and all("position_ids" in k for k in miss.missing_keys), miss
    tok = {t: i for i, t in enumerate(S.BERT_TEST_VOCAB)}
    ids = [tok["[CLS]"]] + [tok.get(ch, tok["[UNK]"]) for ch in BERT_TEXT] + [tok["[SEP]"]]
    with torch.no_grad():
        res = bm(input_ids=torch.tensor([ids]), output_hidden_states=True)
    feat = torch.cat(res["hidden_states"][-3:-2], -1)[0][1:-1]
    np.savez_compressed(os.path.join(GOLD, "bert_large.npz"), feature=feat.numpy().astype(np.float16))
    print("bert", tuple(feat.shape), "rms", float(feat.pow(2).mean().sqrt()))
    # the reference's module needs oracle/ref_import's third-party stubs, which would break the transformers import above
    ref_import.setup()
    from module.mel_processing import spectrogram_torch
    for name, (n, seed) in SPEC_CASES.items():
        y = S.make_waveform(n, seed, sr=32000).unsqueeze(0)
        spec = spectrogram_torch(y, 2048, 32000, 640, 2048, center=False)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), spec=spec.numpy().astype(np.float32))
        print(name, tuple(spec.shape), float(spec.max()))
    import module.mel_processing as mp
    from oracle import mel_filterbank
    mp.librosa_mel_fn = mel_filterbank.mel
    for name, kw, n, seed in (("mel_v3", dict(n_fft=1024, win_size=1024, hop_size=256, num_mels=100, sampling_rate=24000, fmin=0,
                                             fmax=None, center=False), 36000, 3),
                              ("mel_v4", dict(n_fft=1280, win_size=1280, hop_size=320, num_mels=100, sampling_rate=32000, fmin=0,
                                             fmax=None, center=False), 41003, 4)):
        y = S.make_waveform(n, seed, sr=kw["sampling_rate"]).unsqueeze(0)
        m = mp.mel_spectrogram_torch(y, **kw)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), mel=m.numpy().astype(np.float32))
        print(name, tuple(m.shape), float(m.min()), float(m.max()))
    sys.path.insert(0, os.path.join(ref_import.REF_ROOT, "GPT_SoVITS", "eres2net"))
    import kaldi as Kaldi
    from ERes2NetV2 import ERes2NetV2
    em = ERes2NetV2(baseWidth=24, scale=4, expansion=4).eval()
    r = em.load_state_dict(S.make_eres2net_state_dict(seed=0), strict=False)
    assert not r.unexpected_keys and all(k.startswith("seg_1.") for k in r.missing_keys), r      # forward3 never uses the head
    wav = S.make_waveform(SV_N, 5).unsqueeze(0)
    with torch.no_grad():
        feat = torch.stack([Kaldi.fbank(w.unsqueeze(0), num_mel_bins=80, sample_frequency=16000, dither=0) for w in wav])
        emb = em.forward3(feat.clone())
    np.savez_compressed(os.path.join(GOLD, "sv_eres2net.npz"), fbank=feat[0].numpy().astype(np.float32),
                        emb=emb[0].numpy().astype(np.float32))
    print("sv", tuple(feat.shape), tuple(emb.shape), float(emb.abs().mean()))


if __name__ == "__main__":
    main()
