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

# version codes (process_ckpt.py:63-80): "<2-char code>:<symbol version>:<model version>:<lora 0/1>"
_CODES = "00:v1:v1:0 01:v2:v2:0 02:v2:v3:0 03:v2:v3:1 04:v2:v4:1 05:v2:v2Pro:0 06:v2:v2ProPlus:0"
head2version = {c.encode(): [sym, model, lora == "1"] for c, sym, model, lora in (t.split(":") for t in _CODES.split())}
# what the writer stamps (process_ckpt.py:22-27): only these four model versions are written with a code
model_version2byte = {m: c for c, (_, m, lora) in head2version.items()
                      if (m in ("v3", "v4") and lora) or m in ("v2Pro", "v2ProPlus")}
# md5 of the first 8 KiB of the published base models -> model version (process_ckpt.py:81-88); symbol version is v2, no lora
_BASE_MD5 = {"v2": ("dc3c97e17592963677a4a1681f30c653", "6642b37f3dbb1f76882b69937c95a5f3"),
             "v3": ("43797be674a37c1c83ee81081941ed0f",), "v4": ("4f26b9476d0c5033e04162c486074374",),
             "v2Pro": ("c7e9fce2223f3db685cdfa1e6368728a",), "v2ProPlus": ("66b313e39455b57ab1b0bc0b239c9d0a",)}
hash_pretrained_dict = {h: ["v2", m, False] for m, hs in _BASE_MD5.items() for h in hs}


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
    # plain zip archive (oldest format): v1 files are just under 82978 KiB, v2 a little above, v3 is ~750 MB
    kib = os.path.getsize(sovits_path) / 1024
    if kib >= 700 * 1024:
        return "v2", "v3", False
    v = "v1" if kib < 82978 else "v2"
    return v, v, False


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


_PEFT_PREFIX = "cfm.base_model.model."


def merge_lora_v3(base_weight: dict, lora_weight: dict, lora_rank: int, lora_alpha: int = None) -> dict:
    """What the reference does with a v3 / v4 LoRA checkpoint (TTS_infer_pack/TTS.py:556-572): load the base model, wrap
    `cfm` with peft's LoraConfig(target_modules=[to_k, to_q, to_v, to_out.0], r=rank, lora_alpha=rank), load the LoRA state
    dict (strict=False) and `merge_and_unload()` -- restated on the state dict, without peft (not installed here; "parity
    unpinned" against the package, the arithmetic is peft's documented merge):
        W' = W + (lora_alpha / r) * lora_B @ lora_A          per targeted Linear, in fp32.
    Keys: the wrapped model names parameters `cfm.base_model.model.<path>.lora_A.default.weight` / `lora_B.default.weight` and
    the frozen original `<path>.base_layer.weight`; anything else in the LoRA file that names a base parameter overrides it."""
    alpha = lora_rank if lora_alpha is None else lora_alpha
    out = dict(base_weight)
    merged = 0
    for k, a in lora_weight.items():
        if k.endswith(".lora_A.default.weight") or k.endswith(".lora_A.weight"):
            stem = k[:k.index(".lora_A.")]
            kb = k.replace(".lora_A.", ".lora_B.")
            if kb not in lora_weight:
                raise KeyError(f"{k} has no matching lora_B")
            b = lora_weight[kb]
            if a.shape[0] != lora_rank or b.shape[1] != lora_rank:
                raise ValueError(f"{k}: rank {a.shape[0]} / {b.shape[1]} does not match lora_rank {lora_rank}")
            path = stem[len(_PEFT_PREFIX):] if stem.startswith(_PEFT_PREFIX) else stem
            target = ("cfm." + path if stem.startswith(_PEFT_PREFIX) else path) + ".weight"
            if target not in out:
                raise KeyError(f"LoRA target {target} is not a parameter of the base model")
            w = out[target]
            out[target] = (w.float() + (alpha / lora_rank) * (b.float() @ a.float())).to(w.dtype)
            merged += 1
        elif ".lora_B." in k:
            continue
        else:
            kk = k
            if kk.startswith(_PEFT_PREFIX):
                kk = "cfm." + kk[len(_PEFT_PREFIX):]
            kk = kk.replace(".base_layer.", ".")
            if kk in out:
                out[kk] = lora_weight[k]
    if merged == 0:
        raise ValueError("no lora_A / lora_B pairs in the LoRA checkpoint")
    return out
