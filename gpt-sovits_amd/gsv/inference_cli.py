"""Mirror of the reference CLI `GPT_SoVITS/inference_cli.py` (flags :100-122, `synthesize` :11-24,
writes `<output_path>/output.wav` :95-96) on the HIP engines.

The reference CLI goes through `inference_webui.get_tts_wav`: a single-utterance loop that uses the
*naive* AR entry point (EOS masked for the first 11 steps, t2s_model.py:888-889), decodes each sentence
separately, peak-normalises, appends 0.3 s of zeros and scales by 32767 (inference_webui.py:977-1001).
`get_tts_wav` below restates that glue.  Text -> phonemes/BERT and reference audio -> HuBERT tokens /
spectrogram are outside this build: they enter through `frontend`, an object with
    frontend.text_to_segments(text, language)   -> list of {"phones", "bert_features", "norm_text"}
    frontend.reference(ref_audio_path, vits_model) -> (prompt_semantic LongTensor[P], refer_spec [1,1025,Tr])
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


def synthesize(GPT_model_path, SoVITS_model_path, ref_audio_path, ref_text_path, ref_language, target_text_path,
               target_language, output_path, bert_path=None, cnhubert_base_path=None, gpu_number="0", is_half=True,
               frontend=None):
    if frontend is None:
        raise NotImplementedError(
            "the text (G2P/BERT) and reference-audio (HuBERT/STFT) front-ends are outside this build's scope; pass "
            "`frontend` with text_to_segments() and reference() (see the module docstring)")
    with open(ref_text_path, "r", encoding="utf-8") as f:
        ref_text = f.read()
    with open(target_text_path, "r", encoding="utf-8") as f:
        target_text = f.read()
    device = f"cuda:{int(gpu_number)}"
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


def main():
    p = argparse.ArgumentParser(description="GPT-SoVITS Command Line Tool")
    p.add_argument("--gpt_model", required=True, help="Path to the GPT model file")
    p.add_argument("--sovits_model", required=True, help="Path to the SoVITS model file")
    p.add_argument("--ref_audio", required=True, help="Path to the reference audio file")
    p.add_argument("--ref_text", required=True, help="Path to the reference text file")
    p.add_argument("--ref_language", required=True, choices=["中文", "英文", "日文"], help="Language of the reference audio")
    p.add_argument("--target_text", required=True, help="Path to the target text file")
    p.add_argument("--target_language", required=True, choices=["中文", "英文", "日文", "中英混合", "日英混合", "多语种混合"],
                   help="Language of the target text")
    p.add_argument("--output_path", required=True, help="Path to the output directory")
    p.add_argument("--bert_path", default=None)
    p.add_argument("--cnhubert_base_path", default=None)
    p.add_argument("--gpu_number", default="0")
    p.add_argument("--is_half", default="True")
    a = p.parse_args()
    synthesize(a.gpt_model, a.sovits_model, a.ref_audio, a.ref_text, a.ref_language, a.target_text, a.target_language,
               a.output_path, a.bert_path, a.cnhubert_base_path, a.gpu_number, str(a.is_half).lower() in ("true", "1"))


if __name__ == "__main__":
    main()
