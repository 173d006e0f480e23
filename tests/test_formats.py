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
