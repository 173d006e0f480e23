"""TEST INFRASTRUCTURE ONLY -- fixtures for the N3 data formats, produced by the REFERENCE's own functions
(GPT_SoVITS/process_ckpt.py my_save2 / get_sovits_version_from_path_fast / load_sovits_new, api_v2.py's
wave_header_chunk restated with the same stdlib `wave` calls since api_v2 itself needs soundfile/fastapi state).
Run in the build container only:  python oracle/gen_golden_formats.py
Outputs: tests/golden/fmt_<version>.pth (tiny weight files written by the reference's writer),
tests/golden/fmt_expect.json (the reference's version probe per file, md5 of each file, wav header hex)."""
import hashlib
import json
import os
import sys
import wave
from collections import OrderedDict
from io import BytesIO

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, "/root/reference/GPT_SoVITS")
sys.path.insert(0, "/root/reference")


def payload(version):
    g = torch.Generator().manual_seed(7)
    w = OrderedDict()
    w["enc_p.ssl_proj.weight"] = torch.randn(4, 3, 1, generator=g).half()
    w["dec.conv_pre.bias"] = torch.randn(5, generator=g).half()
    return OrderedDict(weight=w, config={"model": {"version": version}, "data": {"sampling_rate": 32000}},
                       info="1epoch_2iteration")


def main():
    import process_ckpt as rp
    cwd = os.getcwd()
    expect = {"files": {}, "wav_header": {},
              "tables": {"head2version": {k.decode(): v for k, v in rp.head2version.items()},
                         "model_version2byte": {k: v.decode() for k, v in rp.model_version2byte.items()},
                         "hash_pretrained_dict": rp.hash_pretrained_dict}}
    for version in ("v3", "v4", "v2Pro", "v2ProPlus"):
        path = os.path.join(GOLD, f"fmt_{version}.pth")
        rp.my_save2(payload(version), path, version)
        probe = rp.get_sovits_version_from_path_fast(path)
        back = rp.load_sovits_new(path)
        assert torch.equal(back["weight"]["dec.conv_pre.bias"], payload(version)["weight"]["dec.conv_pre.bias"])
        expect["files"][f"fmt_{version}.pth"] = {"probe": list(probe), "md5": hashlib.md5(open(path, "rb").read()).hexdigest(),
                                                 "head": open(path, "rb").read(2).decode()}
    # a plain zip (old format): probe falls through to the size rule
    path = os.path.join(GOLD, "fmt_plain.pth")
    torch.save(payload("v2"), path)
    expect["files"]["fmt_plain.pth"] = {"probe": list(rp.get_sovits_version_from_path_fast(path)),
                                        "md5": hashlib.md5(open(path, "rb").read()).hexdigest(), "head": "PK"}
    # api_v2.py:237-249 wave_header_chunk: the same stdlib calls
    for tag, (frames, ch, sw, sr) in {"empty_32k": (b"", 1, 2, 32000), "8bytes_48k": (bytes(range(8)), 1, 2, 48000)}.items():
        buf = BytesIO()
        with wave.open(buf, "wb") as v:
            v.setnchannels(ch)
            v.setsampwidth(sw)
            v.setframerate(sr)
            v.writeframes(frames)
        expect["wav_header"][tag] = {"frames_hex": frames.hex(), "channels": ch, "sample_width": sw, "sample_rate": sr,
                                     "bytes_hex": buf.getvalue().hex()}
    json.dump(expect, open(os.path.join(GOLD, "fmt_expect.json"), "w"), indent=1)
    os.chdir(cwd)
    print(json.dumps(expect["files"], indent=1))


if __name__ == "__main__":
    main()
