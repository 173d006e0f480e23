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

import numpy as np
import torch

from .. import _lib

ACT_LOGCLAMP = 10      # csrc/common.h: log(max(u, 1e-5))

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


def _stft_re_im(y: torch.Tensor, n_fft: int, hop_size: int, win_size: int, center: bool):
    """y [1, n] -> (re | im [T][2 * bins] fp32, T, bins, stream handle); the caller holds torch.cuda.device(y.device)"""
    if center:
        raise NotImplementedError("center=True (every caller in the reference passes center=False)")
    if y.dim() != 2 or y.shape[0] != 1:
        raise ValueError(f"expected a [1, n] waveform, got {tuple(y.shape)}")
    dev = y.device
    if dev.type != "cuda":
        raise RuntimeError("gsv spectrograms run on an MI355X (cuda/HIP device) only")
    n = int(y.shape[1])
    pad = int((n_fft - hop_size) / 2)
    if n <= pad:
        raise ValueError(f"waveform of {n} samples is shorter than the reflect padding {pad}")
    T = (n + 2 * pad - n_fft) // hop_size + 1
    bins = n_fft // 2 + 1
    l = _lib.lib()
    _lib.init(dev.index if dev.index is not None else torch.cuda.current_device())
    x = y[0].to(torch.float32).contiguous()
    basis = _dft_basis(n_fft, win_size, dev)
    frames = torch.empty(T, n_fft, dtype=torch.float32, device=dev)
    ri = torch.empty(T, 2 * bins, dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(l.gsv_op_frame(x.data_ptr(), n, n_fft, hop_size, pad, n_fft, T, frames.data_ptr(), _lib.dtype_code(torch.float32), st),
               "gsv_op_frame")
    d = _lib.ConvDesc()
    d.x, d.w, d.y = frames.data_ptr(), basis.data_ptr(), ri.data_ptr()
    d.T_in = d.T_out = T
    d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = n_fft, 2 * bins, 1, 1, 1, 0
    d.scale, d.out_f32 = 1.0, 1
    _lib.check(l.gsv_op_conv1d(C.byref(d), _lib.dtype_code(torch.float32), st), "gsv_op_conv1d (DFT)")
    return ri, T, bins, st


@torch.no_grad()
def spectrogram_torch(y: torch.Tensor, n_fft: int, sampling_rate: int, hop_size: int, win_size: int, center: bool = False) -> torch.Tensor:
    """y [1, n] (device tensor, any float dtype) -> [1, n_fft // 2 + 1, frames] fp32"""
    with torch.cuda.device(y.device if y.device.type == "cuda" else None):
        ri, T, bins, st = _stft_re_im(y, n_fft, hop_size, win_size, center)
        spec = torch.empty(bins, T, dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().gsv_op_magnitude(ri.data_ptr(), T, bins, 1e-8, 0, spec.data_ptr(), st), "gsv_op_magnitude")
    return spec.unsqueeze(0)


# ---- mel spectrogram of the v3 / v4 reference audio (reference module/mel_processing.py:93-143) -------------------------
def _hz_to_mel(f):
    """Slaney scale (librosa.convert.hz_to_mel, htk=False): linear below 1 kHz (200/3 Hz per mel), logarithmic above"""
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def librosa_mel_fn(sr: int, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """The filterbank the reference takes from `librosa.filters.mel` (mel_processing.py:4, 115; librosa 0.10.2 per
    requirements.txt, not installed here): triangles between n_mels + 2 points equally spaced on the Slaney mel scale, each
    scaled by 2 / (its band width in Hz) ("slaney" area normalisation) -> [n_mels][n_fft // 2 + 1] float32.
    Parity unpinned against the package itself; tests pin it on librosa's documented example values."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


_mel_basis: Dict[Tuple, torch.Tensor] = {}


@torch.no_grad()
def mel_spectrogram_torch(y: torch.Tensor, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int, fmin, fmax,
                          center: bool = False) -> torch.Tensor:
    """y [1, n] -> log(clamp(mel_basis @ |STFT|, 1e-5)) [1, num_mels, frames] fp32: the DFT GEMM, a frame-major magnitude, and
    the filterbank as a second fp32 GEMM with the log-clamp in its epilogue."""
    dev = y.device
    with torch.cuda.device(dev if dev.type == "cuda" else None):
        ri, T, bins, st = _stft_re_im(y, n_fft, hop_size, win_size, center)
        ld = (bins + 7) // 8 * 8
        key = (n_fft, num_mels, sampling_rate, fmin, fmax, str(dev))
        if key not in _mel_basis:
            mb = torch.zeros(num_mels, ld, dtype=torch.float32)
            mb[:, :bins] = torch.from_numpy(librosa_mel_fn(sampling_rate, n_fft, num_mels, fmin, fmax))
            _mel_basis[key] = mb.to(dev).contiguous()
        mag = torch.empty(T, ld, dtype=torch.float32, device=dev)
        out = torch.empty(T, num_mels, dtype=torch.float32, device=dev)
        l = _lib.lib()
        _lib.check(l.gsv_op_magnitude(ri.data_ptr(), T, bins, 1e-8, ld, mag.data_ptr(), st), "gsv_op_magnitude")
        d = _lib.ConvDesc()
        d.x, d.w, d.y = mag.data_ptr(), _mel_basis[key].data_ptr(), out.data_ptr()
        d.T_in = d.T_out = T
        d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = ld, num_mels, 1, 1, 1, 0
        d.scale, d.out_f32, d.post_act = 1.0, 1, ACT_LOGCLAMP
        _lib.check(l.gsv_op_conv1d(C.byref(d), _lib.dtype_code(torch.float32), st), "gsv_op_conv1d (mel)")
    return out.t().unsqueeze(0)
