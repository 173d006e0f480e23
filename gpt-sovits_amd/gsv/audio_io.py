"""Reference-audio file reading and resampling for `TTS.set_ref_audio` (reference TTS_infer_pack/TTS.py:751-819).

The reference reads with torchaudio.load / librosa.load (soundfile, ffmpeg) and resamples with
torchaudio.transforms.Resample / librosa's soxr: none of them exist in this image.  Here: PCM / float WAV through the
standard library's `wave` module + numpy, and polyphase resampling with scipy.signal.resample_poly (Kaiser-windowed FIR).
PARITY UNPINNED: the resampled waveform differs from the reference's resamplers by their filter designs (a few 1e-3 relative);
everything downstream of the waveform is pinned.
"""
from __future__ import annotations

import wave
from math import gcd
from typing import Tuple

import numpy as np


def load_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, n] in [-1, 1), sample rate)"""
    with wave.open(path, "rb") as f:
        ch, sw, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        a = v.astype(np.float32) / 8388608.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"{path}: unsupported sample width {sw}")
    return a.reshape(-1, ch).T.copy(), sr


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """x [..., n] float32 -> [..., ceil(n * sr_out / sr_in)]"""
    if sr_in == sr_out:
        return x.astype(np.float32, copy=False)
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(x.astype(np.float64), sr_out // g, sr_in // g, axis=-1).astype(np.float32)
