"""Host-side mirror of the reference's `DiT` estimator (GPT_SoVITS/f5_tts/model/backbones/dit.py:88-194).

It holds the hyper-parameters and owns the HIP engine (include/gsv.h `gsv_cfm_*`, csrc/cfm.hip); the Euler loop that
drives it lives in the library too, behind `gsv.module.models.CFM.inference` -- the estimator is never stepped
from Python.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from .... import _lib


class DiT:
    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100, text_dim=None,
                 conv_layers=0, long_skip_connection=False, device="cuda", dtype=torch.float16):
        if long_skip_connection:
            raise NotImplementedError("long_skip_connection is never enabled by the reference's v3/v4 models (models.py:1219)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the gsv DiT runs on an MI355X (cuda/HIP device) only; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.dtype = dtype
        self.dim, self.depth, self.heads, self.dim_head, self.ff_mult = dim, depth, heads, dim_head, ff_mult
        self.mel_dim = mel_dim
        self.text_dim = mel_dim if text_dim is None else text_dim
        self.conv_layers = conv_layers
        cfg = _lib.DitConfig(dim, depth, heads, dim_head, ff_mult, mel_dim, self.text_dim, conv_layers)
        with torch.cuda.device(self.device):
            _lib.init(idx)
            h = C.c_void_p()
            _lib.check(_lib.lib().gsv_cfm_create(C.byref(cfg), _lib.dtype_code(dtype), C.byref(h)), "gsv_cfm_create")
            self._h = h
            self.stream = torch.cuda.Stream(device=self.device)
        self._loaded = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().gsv_cfm_destroy(h)
            except Exception:
                pass
            self._h = None

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """keys as in the reference DiT, with or without the `cfm.estimator.` / `estimator.` prefix of a SoVITS checkpoint"""
        l = _lib.lib()
        with torch.cuda.device(self.device):
            for k, v in state_dict.items():
                for pre in ("cfm.estimator.", "estimator."):
                    if k.startswith(pre):
                        k = k[len(pre):]
                if not torch.is_tensor(v) or k.startswith("rotary_embed") or k.startswith("text_embed.freqs_cis"):
                    continue            # buffers the library rebuilds
                t = v.detach().to("cpu", torch.float32).contiguous()
                _lib.check(l.gsv_cfm_load_tensor(self._h, k.encode(), t.data_ptr(), t.numel()), f"load {k}")
            _lib.check(l.gsv_cfm_finalize(self._h), "gsv_cfm_finalize")
        self._loaded = True
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
