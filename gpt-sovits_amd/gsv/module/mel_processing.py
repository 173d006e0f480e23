"""`spectrogram_torch` (reference GPT_SoVITS/module/mel_processing.py:40-74) on the HIP library: reflect padding
(n_fft - hop) / 2 on both sides, frames of n_fft samples every hop (center=False), periodic Hann window, one-sided DFT,
magnitude sqrt(re^2 + im^2 + 1e-8) -> [1, n_fft / 2 + 1, frames].

The DFT is one fp32 GEMM on the matrix cores (frames [T][n_fft] x windowed basis [2 * bins][n_fft], exact fp32 products,
`gsv_op_conv1d`); framing and magnitude are `gsv_op_frame` / `gsv_op_magnitude`.  No torch compute fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Tuple

import torch

from .. import _lib

_basis: Dict[Tuple, torch.Tensor] = {}


def _dft_basis(n_fft: int, win_size: int, device) -> torch.Tensor:
    """[2 * bins][n_fft] fp32: rows 0..bins-1 = w[n] cos(2 pi k n / N), rows bins.. = -w[n] sin(2 pi k n / N); torch.stft pads a
    shorter window to n_fft centred (mel_processing.py passes win_size == n_fft for every reference config)."""
    key = (n_fft, win_size, str(device))
    if key not in _basis:
        bins = n_fft // 2 + 1
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(bins, dtype=torch.float64).unsqueeze(1)
        w = torch.hann_window(win_size, periodic=True, dtype=torch.float64)
        if win_size < n_fft:
            left = (n_fft - win_size) // 2
            w = torch.nn.functional.pad(w, (left, n_fft - win_size - left))
        # the angle is reduced mod N in integers first: cos / sin arguments stay in [0, 2 pi)
        ang = 2.0 * math.pi * ((k.long() * n.long()) % n_fft).double() / n_fft
        basis = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], 0).float()
        _basis[key] = basis.to(device).contiguous()
    return _basis[key]


@torch.no_grad()
def spectrogram_torch(y: torch.Tensor, n_fft: int, sampling_rate: int, hop_size: int, win_size: int, center: bool = False) -> torch.Tensor:
    """y [1, n] (device tensor, any float dtype) -> [1, n_fft // 2 + 1, frames] fp32"""
    if center:
        raise NotImplementedError("center=True (every caller in the reference passes center=False)")
    if y.dim() != 2 or y.shape[0] != 1:
        raise ValueError(f"expected a [1, n] waveform, got {tuple(y.shape)}")
    dev = y.device
    if dev.type != "cuda":
        raise RuntimeError("gsv spectrogram_torch runs on an MI355X (cuda/HIP device) only")
    n = int(y.shape[1])
    pad = int((n_fft - hop_size) / 2)
    if n <= pad:
        raise ValueError(f"waveform of {n} samples is shorter than the reflect padding {pad}")
    T = (n + 2 * pad - n_fft) // hop_size + 1
    bins = n_fft // 2 + 1
    l = _lib.lib()
    with torch.cuda.device(dev):
        _lib.init(dev.index if dev.index is not None else torch.cuda.current_device())
        x = y[0].to(torch.float32).contiguous()
        basis = _dft_basis(n_fft, win_size, dev)
        frames = torch.empty(T, n_fft, dtype=torch.float32, device=dev)
        ri = torch.empty(T, 2 * bins, dtype=torch.float32, device=dev)
        spec = torch.empty(bins, T, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(l.gsv_op_frame(x.data_ptr(), n, n_fft, hop_size, pad, n_fft, T, frames.data_ptr(), _lib.dtype_code(torch.float32), st),
                   "gsv_op_frame")
        d = _lib.ConvDesc()
        d.x, d.w, d.y = frames.data_ptr(), basis.data_ptr(), ri.data_ptr()
        d.T_in = d.T_out = T
        d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = n_fft, 2 * bins, 1, 1, 1, 0
        d.scale, d.out_f32 = 1.0, 1
        _lib.check(l.gsv_op_conv1d(C.byref(d), _lib.dtype_code(torch.float32), st), "gsv_op_conv1d (DFT)")
        _lib.check(l.gsv_op_magnitude(ri.data_ptr(), T, bins, 1e-8, spec.data_ptr(), st), "gsv_op_magnitude")
    return spec.unsqueeze(0)
