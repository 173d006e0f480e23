"""Audio wire formats of the reference's HTTP front-end (api_v2.py:176-249): what leaves the path as bytes.
`raw` and `wav` are byte-level formats and are built here; `ogg` / `aac` need libsndfile / an ffmpeg subprocess in the
reference (api_v2.py:176-179, 193-220) -- third-party encoders, out of scope, refused loudly."""
from __future__ import annotations

import struct
from io import BytesIO

import numpy as np


def pack_raw(io_buffer: BytesIO, data: np.ndarray, rate: int) -> BytesIO:
    """api_v2.py:182-184: the samples' bytes as they are (int16 little endian from the path)."""
    io_buffer.write(data.tobytes())
    return io_buffer


def _riff_pcm16_header(n_bytes: int, channels: int, sample_width: int, rate: int) -> bytes:
    return (b"RIFF" + struct.pack("<I", 36 + n_bytes) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIH H", 16, 1, channels, rate, rate * channels * sample_width, channels * sample_width,
                        8 * sample_width) + b"data" + struct.pack("<I", n_bytes))


def pack_wav(io_buffer: BytesIO, data: np.ndarray, rate: int) -> BytesIO:
    """api_v2.py:187-190 (`soundfile.write(..., format="wav")` of int16 mono data = canonical 44-byte RIFF/WAVE PCM_16
    header + samples).  Like the reference, a NEW buffer is returned and the one passed in is ignored.
    Parity unpinned: soundfile is not installed here, the layout is checked against the stdlib `wave` writer instead."""
    if data.dtype != np.int16:
        raise TypeError("pack_wav expects the path's int16 samples")
    out = BytesIO()
    body = np.ascontiguousarray(data).astype("<i2", copy=False).tobytes()
    out.write(_riff_pcm16_header(len(body), 1, 2, rate))
    out.write(body)
    return out


def pack_audio(io_buffer: BytesIO, data: np.ndarray, rate: int, media_type: str) -> BytesIO:
    """api_v2.py:223-233: dispatch on media type, anything unknown is raw; the buffer is rewound."""
    if media_type in ("ogg", "aac"):
        raise NotImplementedError(f"media_type {media_type!r} needs an external encoder (libsndfile / ffmpeg); "
                                  "request 'wav' or 'raw' and transcode downstream")
    io_buffer = pack_wav(io_buffer, data, rate) if media_type == "wav" else pack_raw(io_buffer, data, rate)
    io_buffer.seek(0)
    return io_buffer


def wave_header_chunk(frame_input: bytes = b"", channels: int = 1, sample_width: int = 2, sample_rate: int = 32000) -> bytes:
    """api_v2.py:237-249: header of a streaming wav response followed by the first frames (data length = what is known
    at that point, 0 for the bare header); later chunks are sent raw."""
    return _riff_pcm16_header(len(frame_input), channels, sample_width, sample_rate) + frame_input
