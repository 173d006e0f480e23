"""`fbank` (reference GPT_SoVITS/eres2net/kaldi.py:519-676, torchaudio's Kaldi-compatible filterbank) on the HIP library, for the
one configuration the reference uses (sv.py:29): num_mel_bins=80, sample_frequency=16000, dither=0, every other option at its
default -- 25 ms frames every 10 ms (snip_edges), DC-offset removal, pre-emphasis 0.97, Povey window, zero padding to 512,
power spectrum, HTK-scale triangular banks from 20 Hz to Nyquist, log(max(., FLT_EPSILON)).

Everything before the power spectrum is linear in the 400 samples of a frame, so it is folded on the host (float64) into the DFT
basis: basis = DFT_512[:, :400] . diag(povey) . P(pre-emphasis) . (I - 11^T / 400).  A frame then costs one row of an fp32 GEMM
(`gsv_op_frame` -> `gsv_op_conv1d`), the power spectrum is `gsv_op_magnitude` (eps < 0) and the mel banks are a second GEMM with
the log-floor in its epilogue.  No torch compute on the path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _lib

ACT_LOG_EPS = 12           # csrc/common.h: log(max(u, FLT_EPSILON))
_cache: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def get_mel_banks(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float, high_freq: float) -> np.ndarray:
    """kaldi.py:436-513 without VTLN -> [num_bins][window_length_padded // 2] float64"""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    lo, hi = float(mel(low_freq)), float(mel(high_freq))
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    m = mel(sample_freq / window_length_padded * np.arange(window_length_padded // 2))[None, :]
    return np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))


def _operators(sr: int, num_mel_bins: int, low_freq: float, high_freq: float, device):
    key = (sr, num_mel_bins, low_freq, high_freq, str(device))
    if key not in _cache:
        win, padded = int(sr * 0.025), 512
        while padded < win:
            padded *= 2
        bins = padded // 2 + 1
        n = np.arange(win, dtype=np.float64)
        povey = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / (win - 1))) ** 0.85            # torch.hann_window(periodic=False) ** 0.85
        pre = np.eye(win) - 0.97 * np.eye(win, k=-1)
        pre[0, 0] = 1.0 - 0.97                                                        # x[0] -= 0.97 * x[0] (replicate padding)
        dc = np.eye(win) - np.full((win, win), 1.0 / win)
        k = np.arange(bins, dtype=np.int64)[:, None]
        ang = 2.0 * math.pi * ((k * np.arange(win, dtype=np.int64)[None, :]) % padded) / padded
        front = (povey[:, None] * pre) @ dc                                           # [win][win]: frame -> windowed frame
        basis = np.concatenate([np.cos(ang) @ front, -np.sin(ang) @ front], 0)        # [2 * bins][win]
        ld = (bins + 7) // 8 * 8
        banks = np.zeros((num_mel_bins, ld))
        banks[:, :padded // 2] = get_mel_banks(num_mel_bins, padded, float(sr), low_freq, high_freq)
        _cache[key] = (torch.from_numpy(basis.astype(np.float32)).to(device).contiguous(),
                       torch.from_numpy(banks.astype(np.float32)).to(device).contiguous())
    return _cache[key]


@torch.no_grad()
def fbank(waveform: torch.Tensor, num_mel_bins: int = 23, sample_frequency: float = 16000.0, dither: float = 0.0,
          low_freq: float = 20.0, high_freq: float = 0.0, **other) -> torch.Tensor:
    """waveform [1, n] device tensor -> [m, num_mel_bins] fp32, m = 1 + (n - 400) // 160"""
    if other or dither != 0.0:
        raise NotImplementedError(f"only the reference's call (sv.py:29) is built: dither=0 and default options, got {other or dither}")
    if waveform.dim() != 2 or waveform.shape[0] != 1:
        raise ValueError(f"expected a [1, n] waveform, got {tuple(waveform.shape)}")
    dev = waveform.device
    if dev.type != "cuda":
        raise RuntimeError("gsv fbank runs on an MI355X (cuda/HIP device) only")
    sr = int(sample_frequency)
    win, shift = int(sr * 0.025), int(sr * 0.010)
    n = int(waveform.shape[1])
    if n < win:
        return torch.empty(0, num_mel_bins, dtype=torch.float32, device=dev)           # kaldi.py:65-66
    m = 1 + (n - win) // shift
    l = _lib.lib()
    with torch.cuda.device(dev):
        _lib.init(dev.index if dev.index is not None else torch.cuda.current_device())
        basis, banks = _operators(sr, num_mel_bins, float(low_freq), float(high_freq), dev)
        bins, ld = basis.shape[0] // 2, banks.shape[1]
        x = waveform[0].to(torch.float32).contiguous()
        frames = torch.empty(m, win, dtype=torch.float32, device=dev)
        ri = torch.empty(m, 2 * bins, dtype=torch.float32, device=dev)
        power = torch.empty(m, ld, dtype=torch.float32, device=dev)
        out = torch.empty(m, num_mel_bins, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        f32 = _lib.dtype_code(torch.float32)
        _lib.check(l.gsv_op_frame(x.data_ptr(), n, win, shift, 0, win, m, frames.data_ptr(), f32, st), "gsv_op_frame")
        d = _lib.ConvDesc()
        d.x, d.w, d.y = frames.data_ptr(), basis.data_ptr(), ri.data_ptr()
        d.T_in = d.T_out = m
        d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = win, 2 * bins, 1, 1, 1, 0
        d.scale, d.out_f32 = 1.0, 1
        _lib.check(l.gsv_op_conv1d(C.byref(d), f32, st), "gsv_op_conv1d (fbank DFT)")
        _lib.check(l.gsv_op_magnitude(ri.data_ptr(), m, bins, -1.0, ld, power.data_ptr(), st), "gsv_op_magnitude")
        d = _lib.ConvDesc()
        d.x, d.w, d.y = power.data_ptr(), banks.data_ptr(), out.data_ptr()
        d.T_in = d.T_out = m
        d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = ld, num_mel_bins, 1, 1, 1, 0
        d.scale, d.out_f32, d.post_act = 1.0, 1, ACT_LOG_EPS
        _lib.check(l.gsv_op_conv1d(C.byref(d), f32, st), "gsv_op_conv1d (mel banks)")
    return out
