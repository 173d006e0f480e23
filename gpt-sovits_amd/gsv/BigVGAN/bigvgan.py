"""Host-side mirror of the reference's `BigVGAN` generator (reference GPT_SoVITS/BigVGAN/bigvgan.py:226-355,
AMPBlock1 :31-131), backed by the HIP vocoder engine with the anti-aliased snake kernel."""
from __future__ import annotations

import torch

from ..module.vocoder import _VocoderEngine


class BigVGAN:
    def __init__(self, h, use_cuda_kernel: bool = False, device="cuda:0", dtype=torch.float16):
        get = h.get if hasattr(h, "get") else (lambda k, d=None: getattr(h, k, d))
        if str(get("resblock", "1")) != "1":
            raise NotImplementedError("only AMPBlock1 (resblock '1') is built")
        act = get("activation", "snakebeta")
        if act not in ("snake", "snakebeta"):
            raise NotImplementedError(f"activation {act}")
        self.h = h
        self._e = _VocoderEngine(1, get("num_mels", 100), get("upsample_initial_channel"), get("upsample_rates"),
                                 get("upsample_kernel_sizes"), get("resblock_kernel_sizes"),
                                 get("resblock_dilation_sizes"), bias_at_final=get("use_bias_at_final", True),
                                 tanh_at_final=get("use_tanh_at_final", True),
                                 snake_logscale=get("snake_logscale", False), device=device, dtype=dtype)

    def load_state_dict(self, sd, strict=True):
        if "generator" in sd and not torch.is_tensor(sd["generator"]):
            sd = sd["generator"]          # bigvgan_generator.pt layout (bigvgan.py:417-459)
        self._e.load_state_dict(sd, strict)
        return self

    def remove_weight_norm(self):
        return self

    def eval(self):
        return self

    def __call__(self, x):
        return self._e(x)
