"""Audio wire formats and response framing of the reference's HTTP front-end (api_v2.py:176-249, 300-373): what leaves the path
as bytes, and in which pieces.

* `raw` and `wav` are byte-level formats and are built here.  `aac` is, as in the reference (api_v2.py:193-220), the output of
  an `ffmpeg` subprocess fed with s16le PCM: used when an ffmpeg executable is on PATH, refused loudly otherwise.  `ogg` needs
  libsndfile through the `soundfile` package (api_v2.py:176-179), which is not installed: refused loudly.
* `streaming_generator` is the chunked-response framing (api_v2.py:346-354): for "wav" one header-only RIFF chunk first, then
  every fragment as raw int16 bytes; other media types are packed fragment by fragment.  It consumes `TTS.run(...)` with
  `return_fragment=True` -- or `ShardedSynthesizer.run_stream` on a multi-GPU node -- and is what BASELINE configs[4] streams.
* `tts_handle` restates api_v2.py:300-373 without the web framework: request dict -> (status, media type, bytes or iterator
  of bytes); `create_app` wraps it in the reference's /tts GET / POST routes when FastAPI is importable (the HTTP server
  itself is outside the hot-path scope: this is the seam a deployment binds to)."""
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
    if media_type == "ogg":
        raise NotImplementedError("media_type 'ogg' needs libsndfile (the `soundfile` package, api_v2.py:176-179), which is not "
                                  "installed; request 'wav', 'raw' or 'aac'")
    if media_type == "aac":
        io_buffer = pack_aac(io_buffer, data, rate)
    elif media_type == "wav":
        io_buffer = pack_wav(io_buffer, data, rate)
    else:
        io_buffer = pack_raw(io_buffer, data, rate)
    io_buffer.seek(0)
    return io_buffer


def pack_aac(io_buffer: BytesIO, data: np.ndarray, rate: int) -> BytesIO:
    """api_v2.py:193-220: ADTS AAC at 192 kb/s from an ffmpeg subprocess fed with the s16le samples."""
    import shutil
    import subprocess
    exe = shutil.which("ffmpeg")
    if exe is None:
        raise NotImplementedError("media_type 'aac' needs an `ffmpeg` executable on PATH (api_v2.py:193-220)")
    p = subprocess.Popen([exe, "-f", "s16le", "-ar", str(rate), "-ac", "1", "-i", "pipe:0", "-c:a", "aac", "-b:a", "192k", "-vn",
                          "-f", "adts", "pipe:1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out, _ = p.communicate(input=np.ascontiguousarray(data).tobytes())
    io_buffer.write(out)
    return io_buffer


def wave_header_chunk(frame_input: bytes = b"", channels: int = 1, sample_width: int = 2, sample_rate: int = 32000) -> bytes:
    """api_v2.py:237-249: header of a streaming wav response followed by the first frames (data length = what is known
    at that point, 0 for the bare header); later chunks are sent raw."""
    return _riff_pcm16_header(len(frame_input), channels, sample_width, sample_rate) + frame_input


def streaming_generator(tts_generator, media_type: str):
    """api_v2.py:346-354: chunked-response framing.  `tts_generator` yields (sample rate, int16 fragment)."""
    first = True
    for sr, chunk in tts_generator:
        if first and media_type == "wav":
            yield wave_header_chunk(sample_rate=sr)
            media_type = "raw"
            first = False
        yield pack_audio(BytesIO(), chunk, sr, media_type).getvalue()


def tts_handle(tts_pipeline, req: dict):
    """api_v2.py:300-373 without the web framework: -> (status code, media type, payload); payload is bytes, an iterator of
    bytes (streaming_mode) or, on failure, the reference's error dict."""
    streaming_mode = req.get("streaming_mode", False)
    media_type = req.get("media_type", "wav")
    if streaming_mode or req.get("return_fragment", False):
        req = dict(req, return_fragment=True)
    try:
        gen = tts_pipeline.run(req)
        if streaming_mode:
            return 200, f"audio/{media_type}", streaming_generator(gen, media_type)
        sr, audio = next(gen)
        return 200, f"audio/{media_type}", pack_audio(BytesIO(), audio, sr, media_type).getvalue()
    except Exception as e:                                              # noqa: BLE001 -- the reference answers 400 with the message
        return 400, "application/json", {"message": "tts failed", "Exception": str(e)}


def create_app(tts_pipeline):
    """the reference's /tts routes (api_v2.py:416-470) over `tts_handle`; needs fastapi (present in this image)"""
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse, Response, StreamingResponse
    app = FastAPI()

    def respond(req: dict):
        code, mt, payload = tts_handle(tts_pipeline, req)
        if code != 200:
            return JSONResponse(status_code=code, content=payload)
        if isinstance(payload, (bytes, bytearray)):
            return Response(payload, media_type=mt)
        return StreamingResponse(payload, media_type=mt)

    @app.get("/tts")
    async def tts_get(request: Request):
        q = dict(request.query_params)
        for k in ("top_k", "batch_size", "seed", "sample_steps"):
            if k in q:
                q[k] = int(q[k])
        for k in ("top_p", "temperature", "batch_threshold", "speed_factor", "fragment_interval", "repetition_penalty"):
            if k in q:
                q[k] = float(q[k])
        for k in ("split_bucket", "streaming_mode", "parallel_infer", "super_sampling"):
            if k in q:
                q[k] = str(q[k]).lower() in ("1", "true")
        return respond(q)

    @app.post("/tts")
    async def tts_post(request: Request):
        return respond(await request.json())
    return app
