"""CPU: the data formats either side of the path against fixtures written / probed by the
reference's own functions (oracle/gen_golden_formats.py): SoVITS weight files with the 2-byte version code
(process_ckpt.py:30-38, 100-138) and the wav / raw audio framing (api_v2.py:182-190, 223-249)."""
import hashlib
import json
import os
import wave
from io import BytesIO

import numpy as np
import pytest
import torch

from gsv import process_ckpt as pc
from gsv import wire

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXPECT = json.load(open(os.path.join(GOLD, "fmt_expect.json")))


@pytest.mark.parametrize("name", sorted(EXPECT["files"]))
def test_version_probe_and_reader_match_reference(name):
    path, e = os.path.join(GOLD, name), EXPECT["files"][name]
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == e["md5"]          # the fixture is the reference's file
    assert list(pc.get_sovits_version_from_path_fast(path)) == e["probe"]
    sd = pc.load_sovits_new(path)
    assert set(sd) == {"weight", "config", "info"} and sd["info"] == "1epoch_2iteration"
    assert sd["weight"]["enc_p.ssl_proj.weight"].dtype == torch.float16 and sd["weight"]["dec.conv_pre.bias"].shape == (5,)
    assert sd["config"]["data"]["sampling_rate"] == 32000


@pytest.mark.parametrize("version", ["v3", "v4", "v2Pro", "v2ProPlus"])
def test_writer_is_byte_identical_to_reference(version, tmp_path):
    ref = os.path.join(GOLD, f"fmt_{version}.pth")
    sd = pc.load_sovits_new(ref)
    out = str(tmp_path / "w.pth")
    pc.my_save2(sd, out, version)
    a, b = open(out, "rb").read(), open(ref, "rb").read()
    assert a[:2] == pc.model_version2byte[version] == b[:2]
    back = pc.load_sovits_new(out)
    for k in sd["weight"]:
        assert torch.equal(back["weight"][k], sd["weight"][k])
    assert pc.get_sovits_version_from_path_fast(out) == pc.head2version[a[:2]]


def test_unknown_version_code_and_executable_payload_are_refused(tmp_path):
    p = str(tmp_path / "bad.pth")
    data = open(os.path.join(GOLD, "fmt_v3.pth"), "rb").read()
    open(p, "wb").write(b"99" + data[2:])
    with pytest.raises(KeyError):                     # head2version[b"99"], as in the reference
        pc.get_sovits_version_from_path_fast(p)

    import argparse                                   # a pickled object in `config`: the reference would unpickle it
    q = str(tmp_path / "cls.pth")
    torch.save({"weight": {}, "config": argparse.Namespace(a=1)}, q)
    with pytest.raises(Exception):
        pc.load_sovits_new(q)


@pytest.mark.parametrize("tag", sorted(EXPECT["wav_header"]))
def test_wave_header_chunk_matches_stdlib_wave_bytes(tag):
    e = EXPECT["wav_header"][tag]
    got = wire.wave_header_chunk(bytes.fromhex(e["frames_hex"]), e["channels"], e["sample_width"], e["sample_rate"])
    assert got.hex() == e["bytes_hex"]


def test_pack_audio_raw_and_wav():
    pcm = (np.arange(-5, 1000, dtype=np.int32) * 37 % 65536 - 32768).astype(np.int16)
    raw = wire.pack_audio(BytesIO(), pcm, 32000, "raw")
    assert raw.tell() == 0 and raw.read() == pcm.tobytes()
    assert wire.pack_audio(BytesIO(), pcm, 32000, "anything-else").read() == pcm.tobytes()      # api_v2.py:230-231
    wav = wire.pack_audio(BytesIO(), pcm, 48000, "wav")
    assert wav.tell() == 0
    with wave.open(wav, "rb") as r:                   # a standard reader sees mono PCM_16 at the given rate, same samples
        assert (r.getnchannels(), r.getsampwidth(), r.getframerate(), r.getnframes()) == (1, 2, 48000, pcm.size)
        assert r.readframes(pcm.size) == pcm.tobytes()
    assert len(wav.getvalue()) == 44 + pcm.nbytes
    for mt in ("ogg", "aac"):
        with pytest.raises(NotImplementedError):
            wire.pack_audio(BytesIO(), pcm, 32000, mt)
    with pytest.raises(TypeError):
        wire.pack_wav(BytesIO(), pcm.astype(np.float32), 32000)


def test_version_tables_equal_the_reference():
    t = EXPECT["tables"]
    assert {k.decode(): v for k, v in pc.head2version.items()} == t["head2version"]
    assert {k: v.decode() for k, v in pc.model_version2byte.items()} == t["model_version2byte"]
    assert pc.hash_pretrained_dict == t["hash_pretrained_dict"]


def test_streaming_framing_and_tts_handle():
    """api_v2.py:300-373: streaming "wav" = one header-only RIFF chunk, then raw fragments; non-streaming = one packed body;
    failures become the reference's 400 payload; ogg is refused (libsndfile absent), aac needs ffmpeg."""
    import numpy as np
    from gsv import wire

    class Fake:
        def run(self, req):
            assert req.get("return_fragment", False) == bool(req.get("streaming_mode", False))
            if req.get("text") == "boom":
                raise RuntimeError("engine failed")
            for i in range(3 if req.get("return_fragment") else 1):
                yield 32000, np.full(10 + i, i + 1, dtype=np.int16)

    code, mt, it = wire.tts_handle(Fake(), {"text": "x", "streaming_mode": True, "media_type": "wav"})
    chunks = list(it)
    assert code == 200 and mt == "audio/wav" and len(chunks) == 4
    assert chunks[0] == wire.wave_header_chunk(sample_rate=32000) and len(chunks[0]) == 44
    assert chunks[1] == np.full(10, 1, dtype=np.int16).tobytes() and chunks[3] == np.full(12, 3, dtype=np.int16).tobytes()
    code, mt, body = wire.tts_handle(Fake(), {"text": "x", "media_type": "wav"})
    assert code == 200 and body[:4] == b"RIFF" and len(body) == 44 + 20
    code, mt, body = wire.tts_handle(Fake(), {"text": "x", "media_type": "raw", "streaming_mode": True})
    assert [len(c) for c in body] == [20, 22, 24]
    code, mt, body = wire.tts_handle(Fake(), {"text": "boom"})
    assert code == 400 and body == {"message": "tts failed", "Exception": "engine failed"}
    code, mt, body = wire.tts_handle(Fake(), {"text": "x", "media_type": "ogg"})
    assert code == 400 and "libsndfile" in body["Exception"]
    import shutil
    if shutil.which("ffmpeg") is None:
        code, _, body = wire.tts_handle(Fake(), {"text": "x", "media_type": "aac"})
        assert code == 400 and "ffmpeg" in body["Exception"]
    app = wire.create_app(Fake())                        # the reference's routes exist
    assert {r.path for r in app.routes if hasattr(r, "methods")} >= {"/tts"}


def test_lora_merge_arithmetic_and_errors():
    """v3 / v4 LoRA checkpoints (TTS.py:556-572): W' = W + (alpha / r) B A on to_q / to_k / to_v / to_out.0 of every DiT block,
    nothing else touched; peft is not installed, so this pins the restated arithmetic, not the package ("parity unpinned")."""
    from gsv import synthetic as S
    vcfg = S.small_vits_config()
    vcfg["model"]["inter_channels"] = vcfg["model"]["hidden_channels"]
    dit = S.small_dit_config()
    dit["text_dim"] = 512
    base = S.make_vits_v3_state_dict(vcfg, seed=12, dit_cfg=dit)
    lw = S.make_lora_state_dict(base, rank=4, seed=3)
    n_targets = len(lw) // 2
    assert n_targets == 4 * dit["depth"]
    merged = pc.merge_lora_v3(base, lw, 4)
    changed = [k for k in base if not torch.equal(base[k], merged[k])]
    assert len(changed) == n_targets and set(merged) == set(base)
    k = "cfm.estimator.transformer_blocks.0.attn.to_q.weight"
    a = lw["cfm.base_model.model.estimator.transformer_blocks.0.attn.to_q.lora_A.default.weight"].float()
    b = lw["cfm.base_model.model.estimator.transformer_blocks.0.attn.to_q.lora_B.default.weight"].float()
    assert torch.allclose(merged[k].float(), base[k].float() + b @ a, atol=1e-6) and merged[k].dtype == base[k].dtype
    assert torch.allclose(pc.merge_lora_v3(base, lw, 4, lora_alpha=8)[k].float(), base[k].float() + 2 * (b @ a), atol=1e-6)
    with pytest.raises(ValueError):
        pc.merge_lora_v3(base, lw, 8)                                   # rank in the file differs from lora_rank
    with pytest.raises(ValueError):
        pc.merge_lora_v3(base, {"enc_p.ssl_proj.weight": base["enc_p.ssl_proj.weight"]}, 4)     # no LoRA pairs at all
    bad = dict(lw)
    bad["cfm.base_model.model.estimator.nowhere.to_q.lora_A.default.weight"] = a
    bad["cfm.base_model.model.estimator.nowhere.to_q.lora_B.default.weight"] = b
    with pytest.raises(KeyError):
        pc.merge_lora_v3(base, bad, 4)
