"""Mirror of the reference CLI `GPT_SoVITS/inference_cli.py` (flags :100-122, `synthesize` :11-24,
writes `<output_path>/output.wav` :95-96) on the HIP engines.

The reference CLI goes through `inference_webui.get_tts_wav`: a single-utterance loop that uses the
*naive* AR entry point (EOS masked for the first 11 steps, t2s_model.py:888-889), decodes each sentence
separately, peak-normalises, appends 0.3 s of zeros and scales by 32767 (inference_webui.py:977-1001).
`get_tts_wav` below restates that glue.

Front-end (`--frontend`, an addition to the reference's flags: the reference hard-wires its third-party G2P packages):
    symbols              the text files hold phoneme symbols separated by white space (gsv.text.g2p.SymbolG2P): runs the
                         whole pipeline with no G2P package at all
    cmudict:<file>       English by look-up in a CMUdict-format dictionary (gsv.text.g2p.DictG2P; the reference ships
                         text/cmudict.rep in that format)
    pkg.module:factory   factory() returns an object with text_to_segments(text, language) and, optionally,
                         reference(ref_audio_path, vits_model); it may also just call gsv.text.cleaner.register_g2p
The built-in front-end is `BuiltinFrontend`: gsv.TTS_infer_pack.TextPreprocessor for the text and, for the reference audio,
WAV -> 16 kHz -> HuBERT-base (gsv.feature_extractor.cnhubert, weights from --cnhubert_base_path) -> extract_latent, and
gsv.module.mel_processing.spectrogram_torch (reference inference_webui.py:590-640, 806-830).
"""
from __future__ import annotations

import argparse
import os
import wave
from typing import Iterator, Tuple

import numpy as np
import torch

from .AR.models.t2s_model import Text2SemanticDecoder
from .module.models import SynthesizerTrn
from .TTS_infer_pack.TTS import load_sovits_new


def get_tts_wav(t2s: Text2SemanticDecoder, vits: SynthesizerTrn, prompt_semantic: torch.Tensor, refer_spec,
                prompt_segment: dict, segments, top_k: int = 20, top_p: float = 1.0, temperature: float = 1.0,
                speed: float = 1.0, hz: int = 50, max_sec: int = 54, pause_second: float = 0.3,
                sampling_rate: int = 32000, seed: int = 0) -> Iterator[Tuple[int, np.ndarray]]:
    """reference inference_webui.py:751-1001 restated for pre-tokenised input; yields (32000, int16)."""
    dev = t2s.device
    zero = np.zeros(int(sampling_rate * pause_second), dtype=np.float32)
    audio = []
    for si, seg in enumerate(segments):
        phones = list(prompt_segment["phones"]) + list(seg["phones"])
        pb, sb = prompt_segment.get("bert_features"), seg.get("bert_features")
        bert = None
        if pb is not None or sb is not None:
            pb = torch.zeros(1024, len(prompt_segment["phones"])) if pb is None else pb
            sb = torch.zeros(1024, len(seg["phones"])) if sb is None else sb
            bert = torch.cat([pb, sb], 1).unsqueeze(0)
        ids = torch.LongTensor(phones).unsqueeze(0)
        pred, idx = t2s.infer_panel(ids, torch.LongTensor([ids.shape[-1]]), prompt_semantic.view(1, -1).to(dev), bert,
                                    top_k=top_k, top_p=top_p, temperature=temperature, early_stop_num=hz * max_sec,
                                    seed=seed + si)
        pred = pred[:, -idx:].unsqueeze(0) if idx > 0 else pred[:, :0].unsqueeze(0)
        if pred.shape[-1] == 0:
            continue
        wav = vits.decode(pred, torch.LongTensor(seg["phones"]).unsqueeze(0), refer_spec, speed=speed,
                          seed=seed + si)[0, 0].float().cpu().numpy()
        peak = np.abs(wav).max()
        if peak > 1:
            wav = wav / peak
        audio.append(wav)
        audio.append(zero)
    if not audio:
        return
    yield sampling_rate, (np.concatenate(audio, 0) * 32767).astype(np.int16)


# reference inference_webui.py:140-160 (i18n keys are the Chinese labels themselves)
dict_language_v1 = {"中文": "all_zh", "英文": "en", "日文": "all_ja", "中英混合": "zh", "日英混合": "ja", "多语种混合": "auto"}
dict_language_v2 = dict(dict_language_v1, **{"粤语": "all_yue", "韩文": "all_ko", "粤英混合": "yue", "韩英混合": "ko",
                                               "多语种混合(粤语)": "auto_yue"})


class BuiltinFrontend:
    """text -> segments through TextPreprocessor with the registered G2P back-ends; reference audio -> (prompt_semantic,
    refer_spec) through the HIP HuBERT engine and spectrogram."""

    def __init__(self, device, cnhubert_base_path=None, hubert_state_dict=None, bert_fn=None, version="v2", cut="cut0"):
        from .TTS_infer_pack.TextPreprocessor import TextPreprocessor
        self.device, self.version, self.cut = torch.device(device), version, cut
        self.tp = TextPreprocessor(bert_fn=bert_fn, device="cpu")
        self._hubert = None
        self._hubert_src = (cnhubert_base_path, hubert_state_dict)

    def text_to_segments(self, text: str, language: str):
        lang = (dict_language_v1 if self.version == "v1" else dict_language_v2).get(language, language)
        return self.tp.preprocess(text, lang, self.cut, self.version)

    def reference(self, ref_audio_path: str, vits_model):
        from .audio_io import load_wav, resample
        from .feature_extractor.cnhubert import CNHubert
        from .module.mel_processing import spectrogram_torch
        if self._hubert is None:
            base, sd = self._hubert_src
            if base is None and sd is None:
                raise FileNotFoundError("--cnhubert_base_path (directory with the chinese-hubert-base weights) is required")
            self._hubert = CNHubert(base, device=self.device, state_dict=sd)
        raw, sr = load_wav(ref_audio_path)
        mono = raw.mean(0)
        wav16k = resample(mono, sr, 16000)
        if wav16k.shape[0] > 160000 or wav16k.shape[0] < 48000:
            raise OSError("参考音频在3~10秒范围外，请更换！")                 # inference_webui.py:812-814
        zero = np.zeros(int(32000 * 0.3), dtype=np.float32)
        w = torch.from_numpy(np.concatenate([wav16k, zero])).to(self.device)
        ssl = self._hubert.model(w.unsqueeze(0))["last_hidden_state"].transpose(1, 2)
        prompt_semantic = vits_model.extract_latent(ssl)[0, 0]
        audio = torch.from_numpy(resample(mono, sr, 32000)).unsqueeze(0).to(self.device)
        maxx = float(audio.abs().max())
        if maxx > 1:
            audio = audio / min(2.0, maxx)
        spec = spectrogram_torch(audio, 2048, 32000, 640, 2048, center=False)
        return prompt_semantic, spec


def make_frontend(spec: str, device, cnhubert_base_path=None, version="v2"):
    """--frontend value -> front-end object (see the module docstring)"""
    from .text import cleaner, g2p
    if spec == "symbols":
        for lang in ("zh", "ja", "en", "ko", "yue"):
            cleaner.register_g2p(lang, _SymbolAny(lang))
        return BuiltinFrontend(device, cnhubert_base_path, version=version)
    if spec.startswith("cmudict:"):
        cleaner.register_g2p("en", g2p.DictG2P(spec[len("cmudict:"):]))
        return BuiltinFrontend(device, cnhubert_base_path, version=version)
    if ":" in spec:
        import importlib
        mod, fn = spec.split(":", 1)
        obj = getattr(importlib.import_module(mod), fn)()
        if obj is None or not hasattr(obj, "text_to_segments"):
            return BuiltinFrontend(device, cnhubert_base_path, version=version)       # the factory only registered G2P back-ends
        if not hasattr(obj, "reference"):
            obj.reference = BuiltinFrontend(device, cnhubert_base_path, version=version).reference
        return obj
    raise ValueError(f"--frontend {spec!r}: expected 'symbols', 'cmudict:<file>' or 'pkg.module:factory'")


class _SymbolAny:
    """SymbolG2P for any language: zh / yue must also return word2ph (one entry per character of the normalised text)."""

    def __init__(self, lang):
        self.lang = lang

    def text_normalize(self, text):
        return " ".join(text.split())

    def g2p(self, norm):
        ph = norm.split()
        if self.lang not in ("zh", "yue"):
            return ph
        w2p = [0] * len(norm)
        pos = 0
        for tok in ph:                      # every symbol is charged to its first character
            pos = norm.index(tok, pos)
            w2p[pos] += 1
            pos += len(tok)
        return ph, w2p


def synthesize(GPT_model_path, SoVITS_model_path, ref_audio_path, ref_text_path, ref_language, target_text_path,
               target_language, output_path, bert_path=None, cnhubert_base_path=None, gpu_number="0", is_half=True,
               frontend=None):
    if frontend is None:
        raise ValueError("pass `frontend` (an object, or a --frontend string for make_frontend): the reference's G2P "
                         "packages are third-party and not part of this build (see the module docstring)")
    device = f"cuda:{int(gpu_number)}"
    if isinstance(frontend, str):
        frontend = make_frontend(frontend, device, cnhubert_base_path)
    with open(ref_text_path, "r", encoding="utf-8") as f:
        ref_text = f.read()
    with open(target_text_path, "r", encoding="utf-8") as f:
        target_text = f.read()
    dtype = torch.float16 if is_half else torch.float32
    s1 = torch.load(GPT_model_path, map_location="cpu", weights_only=True)
    t2s = Text2SemanticDecoder(s1["config"], device=device, dtype=dtype, max_batch=1, max_seq=2048)
    t2s.load_state_dict(s1["weight"])
    s2 = load_sovits_new(SoVITS_model_path)
    hps = s2["config"]
    mcfg = dict(hps["model"])
    version = mcfg.pop("version", "v2")
    vits = SynthesizerTrn(hps["data"]["filter_length"] // 2 + 1, hps["train"]["segment_size"] // hps["data"]["hop_length"],
                          n_speakers=hps["data"]["n_speakers"], version=version if version in ("v1", "v2") else "v2",
                          device=device, dtype=dtype, **mcfg)
    vits.load_state_dict(s2["weight"])
    prompt_semantic, refer_spec = frontend.reference(ref_audio_path, vits)
    prompt_seg = frontend.text_to_segments(ref_text, ref_language)[0]
    segments = frontend.text_to_segments(target_text, target_language)
    result = list(get_tts_wav(t2s, vits, prompt_semantic, refer_spec, prompt_seg, segments,
                              max_sec=s1["config"]["data"]["max_sec"]))
    if result:
        sr, audio = result[-1]
        os.makedirs(output_path, exist_ok=True)
        out = os.path.join(output_path, "output.wav")
        with wave.open(out, "wb") as w:            # 16-bit PCM, what soundfile.write produces for int16 (inference_cli.py:95-96)
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(sr)
            w.writeframes(audio.tobytes())
        print(f"Audio saved to {out}")


def main(argv=None):
    p = argparse.ArgumentParser(description="GPT-SoVITS Command Line Tool")
    p.add_argument("--gpt_model", required=True, help="Path to the GPT model file")
    p.add_argument("--sovits_model", required=True, help="Path to the SoVITS model file")
    p.add_argument("--ref_audio", required=True, help="Path to the reference audio file")
    p.add_argument("--ref_text", required=True, help="Path to the reference text file")
    p.add_argument("--ref_language", required=True, choices=["中文", "英文", "日文"], help="Language of the reference audio")   # as the reference (:107-108)
    p.add_argument("--target_text", required=True, help="Path to the target text file")
    p.add_argument("--target_language", required=True, choices=["中文", "英文", "日文", "中英混合", "日英混合", "多语种混合"],
                   help="Language of the target text")
    p.add_argument("--output_path", required=True, help="Path to the output directory")
    p.add_argument("--bert_path", default=None)
    p.add_argument("--cnhubert_base_path", default=None)
    p.add_argument("--gpu_number", default="0")
    p.add_argument("--is_half", default="True")
    p.add_argument("--frontend", default=os.environ.get("GSV_FRONTEND", "symbols"),
                   help="text front-end: 'symbols' (text files hold phoneme symbols), 'cmudict:<file>' (English dictionary G2P) "
                        "or 'pkg.module:factory' (default: $GSV_FRONTEND or 'symbols')")
    a = p.parse_args(argv)
    synthesize(a.gpt_model, a.sovits_model, a.ref_audio, a.ref_text, a.ref_language, a.target_text, a.target_language,
               a.output_path, a.bert_path, a.cnhubert_base_path, a.gpu_number, str(a.is_half).lower() in ("true", "1"),
               frontend=a.frontend)


if __name__ == "__main__":
    main()
