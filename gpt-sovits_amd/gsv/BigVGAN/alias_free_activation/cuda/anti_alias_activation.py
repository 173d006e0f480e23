"""Host-side mirror of the reference's only native FFI, the pybind11 module
`anti_alias_activation_cuda` (reference BigVGAN/alias_free_activation/cuda/
anti_alias_activation.cpp:19-22, called from activation1d.py:22-23), backed by the
gfx950 kernel `gsv_aa_act_forward` (include/gsv.h).
"""
from __future__ import annotations

import ctypes as C

import torch

from .... import _lib


def forward(inputs: torch.Tensor, up_ftr: torch.Tensor, down_ftr: torch.Tensor, alpha: torch.Tensor,
            beta: torch.Tensor) -> torch.Tensor:
    """Same contract as the reference's `fwd_cuda` (anti_alias_activation_cuda.cu:212-246):
    inputs [B, C, T]; up_ftr/down_ftr [1, 1, 12]; alpha/beta [C] in LOG scale; all five tensors of
    one dtype on one device; returns a new [B, C, T] tensor; runs on the current stream."""
    if inputs.device.type != "cuda":
        raise RuntimeError("anti_alias_activation.forward needs a HIP device tensor (no CPU path)")
    if inputs.dim() != 3:
        raise ValueError("expected a [B, C, T] tensor")
    dt = inputs.dtype
    for t in (up_ftr, down_ftr, alpha, beta):
        if t.dtype != dt or t.device != inputs.device:
            raise ValueError("all tensors must share dtype and device (as in the reference)")
    B, Cc, T = inputs.shape
    x = inputs.contiguous()
    out = torch.empty_like(x)
    if T == 0:
        return out
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        _lib.init(idx)
        uf, df = up_ftr.contiguous().view(-1), down_ftr.contiguous().view(-1)
        a, b = alpha.contiguous(), beta.contiguous()
        if uf.numel() != 12 or df.numel() != 12 or a.numel() != Cc or b.numel() != Cc:
            raise ValueError("filters must have 12 taps and alpha/beta one value per channel")
        s = C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
        _lib.check(_lib.lib().gsv_aa_act_forward(x.data_ptr(), out.data_ptr(), uf.data_ptr(), df.data_ptr(),
                                                 a.data_ptr(), b.data_ptr(), B, Cc, T, _lib.dtype_code(dt), s),
                   "gsv_aa_act_forward")
    return out
