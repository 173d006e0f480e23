"""GPU: BASELINE configs[0] plumbing end to end -- `python -m gsv.inference_cli` with the reference's flags on synthetic
checkpoint FILES (GPT .ckpt, SoVITS .pth with the 2-byte version header, HuBERT pytorch_model.bin), a WAV reference and text
files, writes <output_path>/output.wav; and `TTS.run` from raw text + ref_audio_path + prompt_text with the built-in front-end."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_wav(path, x, sr):
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


@pytest.fixture(scope="module")
def assets(tmp_path_factory):
    from gsv import synthetic as S
    d = tmp_path_factory.mktemp("cli")
    cfg = {k: dict(v) for k, v in S.T2S_V2_CONFIG.items()}
    cfg["data"]["max_sec"] = 0.6                          # early_stop_num = 50 * 0.6 = 30 tokens
    torch.save({"weight": {"model." + k: v.half() for k, v in S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0, suppress_eos=True).items()},
                "config": cfg, "info": "synthetic"}, d / "gpt.ckpt")
    # v1 / v2 SoVITS files are plain torch.save archives (only v3 / v4-LoRA / v2Pro files carry the 2-byte version code)
    torch.save({"weight": {k: v.half() for k, v in S.make_vits_state_dict(S.VITS_V2_CONFIG, seed=0).items()},
                "config": dict(S.VITS_V2_CONFIG), "info": "synthetic"}, d / "sovits.pth")
    os.makedirs(d / "hubert")
    torch.save({k: v.half() for k, v in S.make_hubert_state_dict(seed=0).items()}, d / "hubert" / "pytorch_model.bin")
    _write_wav(str(d / "ref.wav"), S.make_waveform(4 * 22050, 5, sr=22050).numpy(), 22050)
    (d / "ref.txt").write_text("HH AH0 L OW1 , DH IH1 S IH1 Z AH0 R EH1 F ER0 AH0 N S .", encoding="utf-8")
    (d / "target.txt").write_text("W AH1 N T UW1 TH R IY1 . F AO1 R F AY1 V S IH1 K S !", encoding="utf-8")
    return d


def test_inference_cli_writes_output_wav(assets):
    d = assets
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "gpt-sovits_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "gsv.inference_cli", "--gpt_model", str(d / "gpt.ckpt"), "--sovits_model", str(d / "sovits.pth"),
           "--ref_audio", str(d / "ref.wav"), "--ref_text", str(d / "ref.txt"), "--ref_language", "英文",
           "--target_text", str(d / "target.txt"), "--target_language", "英文", "--output_path", str(d / "out"),
           "--cnhubert_base_path", str(d / "hubert"), "--frontend", "symbols"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Audio saved to" in r.stdout
    with wave.open(str(d / "out" / "output.wav"), "rb") as f:
        assert f.getframerate() == 32000 and f.getnchannels() == 1 and f.getsampwidth() == 2
        n = f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2")
    # one sentence (cut0): 30 tokens x 1280 samples + 0.3 s pause
    assert n == 30 * 1280 + 9600 and int(np.abs(pcm[: 30 * 1280]).max()) > 0 and not pcm[30 * 1280:].any()


def test_tts_run_from_raw_text_and_wav(assets):
    from gsv import synthetic as S
    from gsv.inference_cli import make_frontend
    from gsv.TTS_infer_pack.TTS import TTS
    d = assets
    make_frontend("symbols", "cuda:0")                                    # registers the symbol G2P back-ends
    tts = TTS({"device": "cuda:0", "is_half": True, "version": "v2", "max_batch": 4, "max_seq": 700})
    tts.init_t2s_weights(str(d / "gpt.ckpt"))
    tts.init_vits_weights(str(d / "sovits.pth"))
    tts.init_cnhuhbert_weights(str(d / "hubert"))
    inputs = {"text": (d / "target.txt").read_text(encoding="utf-8"), "text_lang": "en", "ref_audio_path": str(d / "ref.wav"),
              "prompt_text": (d / "ref.txt").read_text(encoding="utf-8"), "prompt_lang": "en", "text_split_method": "cut4",
              "batch_size": 4, "top_k": 5, "seed": 1}
    out = list(tts.run(inputs))
    sr, audio = out[-1]
    assert sr == 32000 and audio.dtype == np.int16
    assert audio.size == 2 * (30 * 1280 + 9600)                           # cut4: two sentences
    assert tts.prompt_cache["ref_audio_path"] == str(d / "ref.wav") and len(tts.prompt_cache["phones"]) > 10
    out2 = list(tts.run(inputs))                                          # cached prompt, same seed: identical audio
    assert np.array_equal(out2[-1][1], audio)
    with pytest.raises(ValueError):
        list(tts.run(dict(inputs, text_lang="xx")))
