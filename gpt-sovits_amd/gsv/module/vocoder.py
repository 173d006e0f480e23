"""Shared host wrapper of the HIP vocoder engine (include/gsv.h `gsv_vocoder_*`), used by the two
reference-named mirrors: `gsv.module.models.Generator` (v4, reference module/models.py:407) and
`gsv.BigVGAN.bigvgan.BigVGAN` (v3, reference BigVGAN/bigvgan.py:226)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import torch

from .. import _lib


class _VocoderEngine:
    def __init__(self, kind: int, in_channels: int, upsample_initial_channel: int, upsample_rates, upsample_kernel_sizes,
                 resblock_kernel_sizes, resblock_dilation_sizes, bias_at_final: bool, tanh_at_final: bool,
                 snake_logscale: bool, device, dtype):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the gsv vocoders run on an MI355X (cuda/HIP device) only; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.dtype = dtype
        self.upsample_rates = list(upsample_rates)
        cfg = _lib.VocoderConfig()
        cfg.kind, cfg.in_channels, cfg.upsample_initial_channel = kind, in_channels, upsample_initial_channel
        cfg.n_ups = len(upsample_rates)
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.up_rates[i], cfg.up_kernels[i] = u, k
        cfg.n_resblocks = len(resblock_kernel_sizes)
        for j, (k, ds) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            cfg.rb_kernels[j] = k
            for c, d in enumerate(ds):
                cfg.rb_dilations[j][c] = d
        cfg.bias_at_final, cfg.tanh_at_final, cfg.snake_logscale = int(bias_at_final), int(tanh_at_final), int(snake_logscale)
        self.in_channels = in_channels
        with torch.cuda.device(self.device):
            _lib.init(idx)
            h = C.c_void_p()
            _lib.check(_lib.lib().gsv_vocoder_create(C.byref(cfg), _lib.dtype_code(dtype), C.byref(h)), "gsv_vocoder_create")
            self._h = h
            self.stream = torch.cuda.Stream(device=self.device)
        self._loaded = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().gsv_vocoder_destroy(h)
            except Exception:
                pass
            self._h = None

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        l = _lib.lib()
        with torch.cuda.device(self.device):
            for k, v in state_dict.items():
                if not torch.is_tensor(v) or "filter" in k or v.numel() == 0:
                    continue            # Kaiser-sinc filters are constants rebuilt in the library
                t = v.detach().to("cpu", torch.float32).contiguous()
                _lib.check(l.gsv_vocoder_load_tensor(self._h, k.encode(), t.data_ptr(), t.numel()), f"load {k}")
            _lib.check(l.gsv_vocoder_finalize(self._h), "gsv_vocoder_finalize")
        self._loaded = True
        return self

    @torch.no_grad()
    def forward(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [1, in_channels, F] -> waveform [1, 1, F * prod(upsample_rates)]"""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if mel.dim() != 3 or mel.shape[0] != 1 or mel.shape[1] != self.in_channels or mel.shape[2] < 1:
            raise ValueError(f"expected mel of shape [1, {self.in_channels}, F>=1], got {tuple(mel.shape)}")
        F_ = int(mel.shape[2])
        with torch.cuda.device(self.device):
            m = mel[0].to(self.device, torch.float32).contiguous()
            wav = torch.empty(F_ * math.prod(self.upsample_rates), dtype=torch.float32, device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # after the conversion above
            _lib.check(_lib.lib().gsv_vocoder_forward(self._h, m.data_ptr(), F_, wav.data_ptr(),
                                                      C.c_void_p(self.stream.cuda_stream)), "gsv_vocoder_forward")
            self.stream.synchronize()
        return wav.to(self.dtype).view(1, 1, -1)

    __call__ = forward
