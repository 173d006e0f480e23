"""SoVITS weight-file format (reference GPT_SoVITS/process_ckpt.py:20-138): a torch zip archive whose first two bytes
(`PK`) may be replaced by a 2-character version code.  Reading puts `PK` back and loads with a NON-executing loader
(`weights_only=True`); the reference unpickles (`weights_only=False`, process_ckpt.py:136-138) -- files whose `config`
entry is a pickled custom class are therefore refused here, loudly, instead of being executed."""
from __future__ import annotations

import hashlib
import io
import os
from typing import List, Tuple

import torch

# process_ckpt.py:72-80: code -> [symbol version, model version, is_lora]
head2version = {
    b"00": ["v1", "v1", False],
    b"01": ["v2", "v2", False],
    b"02": ["v2", "v3", False],
    b"03": ["v2", "v3", True],
    b"04": ["v2", "v4", True],
    b"05": ["v2", "v2Pro", False],
    b"06": ["v2", "v2ProPlus", False],
}
# process_ckpt.py:22-27: what the writer stamps
model_version2byte = {"v3": b"03", "v4": b"04", "v2Pro": b"05", "v2ProPlus": b"06"}
# process_ckpt.py:81-88: md5 of the first 8192 bytes of the published base models
hash_pretrained_dict = {
    "dc3c97e17592963677a4a1681f30c653": ["v2", "v2", False],
    "43797be674a37c1c83ee81081941ed0f": ["v2", "v3", False],
    "6642b37f3dbb1f76882b69937c95a5f3": ["v2", "v2", False],
    "4f26b9476d0c5033e04162c486074374": ["v2", "v4", False],
    "c7e9fce2223f3db685cdfa1e6368728a": ["v2", "v2Pro", False],
    "66b313e39455b57ab1b0bc0b239c9d0a": ["v2", "v2ProPlus", False],
}


def get_hash_from_file(sovits_path: str) -> str:
    """process_ckpt.py:92-97"""
    with open(sovits_path, "rb") as f:
        return hashlib.md5(f.read(8192)).hexdigest()


def get_sovits_version_from_path_fast(sovits_path: str):
    """process_ckpt.py:100-126: (1) known base model by hash, (2) 2-byte version code, (3) plain zip: by file size
    (< 82978 KiB v1, < 700 MiB v2, else v3).  Returns [version, model_version, if_lora_v3] like the reference
    (a list for cases 1-2, a tuple for case 3)."""
    h = get_hash_from_file(sovits_path)
    if h in hash_pretrained_dict:
        return hash_pretrained_dict[h]
    with open(sovits_path, "rb") as f:
        version = f.read(2)
    if version != b"PK":
        return head2version[version]           # unknown code: KeyError, as in the reference
    size = os.path.getsize(sovits_path)
    if size < 82978 * 1024:
        model_version = version = "v1"
    elif size < 700 * 1024 * 1024:
        model_version = version = "v2"
    else:
        version, model_version = "v2", "v3"
    return version, model_version, False


def load_sovits_new(sovits_path: str) -> dict:
    """process_ckpt.py:129-138 with a non-executing loader."""
    with open(sovits_path, "rb") as f:
        data = f.read()
    if data[:2] != b"PK":
        data = b"PK" + data[2:]
    return torch.load(io.BytesIO(data), map_location="cpu", weights_only=True)


def my_save2(fea, path: str, model_version: str) -> None:
    """process_ckpt.py:30-38: torch.save, then overwrite the zip magic with the version code."""
    bio = io.BytesIO()
    torch.save(fea, bio)
    data = bio.getvalue()
    with open(path, "wb") as f:
        f.write(model_version2byte[model_version] + data[2:])
