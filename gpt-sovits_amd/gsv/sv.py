"""`SV` (reference GPT_SoVITS/sv.py:11-32): the speaker-verification embedding of v2Pro / v2ProPlus -- Kaldi fbank (80 bins, 16 kHz)
-> ERes2NetV2(baseWidth=24, scale=4, expansion=4).forward3 -> [B, 20480].  The engine is fp32 whatever `is_half` says (the result
is cast to half like the reference's); weights come from an in-memory state dict or from the reference's checkpoint path loaded
with `weights_only=True` (the file is a plain tensor state dict)."""
from __future__ import annotations

import torch

from .eres2net import kaldi as Kaldi
from .eres2net.ERes2NetV2 import ERes2NetV2

sv_path = "GPT_SoVITS/pretrained_models/sv/pretrained_eres2netv2w24s4ep4.ckpt"


class SV:
    def __init__(self, device, is_half: bool, state_dict=None, path: str = sv_path):
        if state_dict is None:
            state_dict = torch.load(path, map_location="cpu", weights_only=True)
        self.embedding_model = ERes2NetV2(state_dict, device=device, baseWidth=24, scale=4, expansion=4)
        self.device = torch.device(device)
        self.is_half = is_half

    @torch.no_grad()
    def compute_embedding3(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B, n] at 16 kHz -> [B, 20480]"""
        wav = wav.to(self.device)
        feat = torch.stack([Kaldi.fbank(w.unsqueeze(0), num_mel_bins=80, sample_frequency=16000, dither=0) for w in wav])
        emb = self.embedding_model.forward3(feat)
        return emb.half() if self.is_half else emb
