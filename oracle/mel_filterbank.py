"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `librosa.filters.mel` (librosa 0.10.2, the reference's requirements.txt pin;
the package is not installed here), used by the reference at module/mel_processing.py:115 for the v3 / v4 reference mel.

Published algorithm (librosa/filters.py `mel`, librosa/core/convert.py `hz_to_mel` / `mel_to_hz`, htk=False, norm="slaney"):
the Slaney scale is linear below 1 kHz (200/3 Hz per mel) and logarithmic above (step ln(6.4)/27 per mel); n_mels + 2 points
equally spaced on that scale between fmin and fmax delimit n_mels triangles over the FFT bin frequencies; each triangle is
scaled by 2 / (upper edge - lower edge).  Written band by band in scalar loops, on purpose unlike the product's vectorised
version (gsv/module/mel_processing.py), so the two check each other.  Parity unpinned against the package; pinned on the
example values in librosa's documentation (tests/test_text_frontend.py)."""
import math

import numpy as np

F_SP = 200.0 / 3
MIN_LOG_HZ = 1000.0
MIN_LOG_MEL = MIN_LOG_HZ / F_SP
LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f: float) -> float:
    return MIN_LOG_MEL + math.log(f / MIN_LOG_HZ) / LOGSTEP if f >= MIN_LOG_HZ else f / F_SP


def mel_to_hz(m: float) -> float:
    return MIN_LOG_HZ * math.exp(LOGSTEP * (m - MIN_LOG_MEL)) if m >= MIN_LOG_MEL else F_SP * m


def mel_frequencies(n: int, fmin: float, fmax: float):
    lo, hi = hz_to_mel(fmin), hz_to_mel(fmax)
    return [mel_to_hz(lo + (hi - lo) * i / (n - 1)) for i in range(n)]


def mel(sr: int, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax=None) -> np.ndarray:
    fmax = sr / 2.0 if fmax is None else float(fmax)
    bins = 1 + n_fft // 2
    edges = mel_frequencies(n_mels + 2, fmin, fmax)
    out = np.zeros((n_mels, bins), dtype=np.float64)
    for i in range(n_mels):
        lo, mid, hi = edges[i], edges[i + 1], edges[i + 2]
        for k in range(bins):
            f = (sr / 2.0) * k / (bins - 1)
            w = min((f - lo) / (mid - lo), (hi - f) / (hi - mid))
            if w > 0:
                out[i, k] = w * 2.0 / (hi - lo)
    return out.astype(np.float32)


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """reference module/mel_processing.py:93-143 on the CPU: reflect padding (n_fft - hop) / 2, torch.stft with a periodic Hann
    window, sqrt(re^2 + im^2 + 1e-8), filterbank matmul, log(clamp(min=1e-5)).  y [1, n] fp32 -> [1, num_mels, frames]."""
    import torch
    basis = torch.from_numpy(mel(sampling_rate, n_fft, num_mels, fmin, fmax))
    p = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=center,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-8)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))
