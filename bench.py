#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: synthesised audio seconds per wall second (1/RTF)
and p50 utterance latency of the v2 pipeline, whole job over N GPUs.

One "step" = one pass of the hot path over one batch of synthetic utterances per GPU
(BASELINE.json configs[1]: B=32 fixed-length sentences, fp16): TTS.run -> to_batch ->
AR prefill + 100 KV-cached decode steps + sampling -> one time-axis-concatenated SoVITS decode
(enc_p, flow, HiFi-GAN generator) -> peak-normalise / silence / int16.  Weights (155 MB + 82 MB),
prompt cache and reference conditioning are resident in HBM before the timed region; per-step host
inputs are the tokenised segments (a few KB).  N > 1: one process per GPU (torchrun), utterances
scattered from rank 0 and int16 fragments gathered back over RCCL inside the timed step
(gsv/sharding.py); weak scaling (32 utterances per GPU).

`--total-utterances M` (BASELINE configs[2]): rank 0 submits M utterances per step, cut into batches of `--batch` that
the ranks take from a work queue (strong scaling over the job, `"scaling": "strong"`).

Prints ONE JSON line (rank 0).  `roofline` = the kernel with the largest share of GPU time: the persistent AR decode engine
`t2s_mega_kernel` (one launch = all decode steps after step 0 of one batch; HBM-bound): algorithmic
bytes of its steps / its duration from HIP events on the engine's stream (gsv_t2s_decode_info), measured on the LAST timed
step; `roofline_step` / `roofline_prefill` / `roofline_generator` put the other stages against their bounds; `total_1024` /
`total_1024_b128` run BASELINE configs[2]'s 1024 utterances on this one GPU (batches of 32 / of 128 = one persistent AR launch
per batch), `long_form` streams configs[4]'s 140 sentences in reading order (time to the first fragment; `first_batch_8`: the
same with a first batch of 8 sentences), `fp32` repeats a few steps with the parity dtype, `v3` times BASELINE configs[3] (one
934-frame chunk: 32 Euler steps of the DiT + BigVGAN), `cold_prompt` the reference-audio front-end (WAV -> HuBERT -> codes,
spectrogram) that precedes a first utterance.
`cpu_baseline` = the oracle (CPU restatement, kind "port") on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gpt-sovits_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver only supports dmabuf IPC: RCCL between ranks needs this (already exported on the pool; kept here so a
# bare `torchrun bench.py` works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build_tts(device, tokens: int, batch: int, dtype_half: bool = True):
    from gsv import synthetic as S
    from gsv.TTS_infer_pack.TTS import TTS
    t2s_cfg = {k: dict(v) for k, v in S.T2S_V2_CONFIG.items()}
    t2s_cfg["data"]["max_sec"] = tokens / 50.0          # early_stop_num = hz * max_sec = tokens (TTS.py:1224)
    tts = TTS({"device": str(device), "is_half": dtype_half, "version": "v2", "max_batch": batch,
               "max_seq": 80 + 100 + tokens + 16})
    tts.init_t2s_weights(state={"weight": S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0, suppress_eos=True),
                                "config": t2s_cfg})
    vcfg = dict(S.VITS_V2_CONFIG)
    tts.init_vits_weights(state={"weight": S.make_vits_state_dict(S.VITS_V2_CONFIG, seed=0), "config": vcfg})
    return tts


def make_segments(n: int, seed: int = 0):
    from gsv import synthetic as S
    utt = S.make_utterances(n, seed=seed)
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": it["norm_text"]}
            for it in utt["items"]]
    return utt, segs


def cpu_baseline(n_utt: int, tokens: int):
    """oracle (CPU restatement) on the first n_utt utterances of the same synthetic workload."""
    from gsv import synthetic as S
    from oracle.t2s_oracle import T2SOracle
    from oracle.vits_oracle import VitsOracle
    # the box reports every host core but a 1-GPU slot owns a 16-core share; small-tensor torch ops
    # do not scale past a few threads anyway, so the baseline uses 8 (the count BASELINE.md section 2 used)
    cores = min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    utt, segs = make_segments(n_utt)
    sd = S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0, suppress_eos=True)
    vsd = S.make_vits_state_dict(S.VITS_V2_CONFIG, seed=0)
    t2s, vits = T2SOracle(sd, S.T2S_V2_CONFIG), VitsOracle(vsd, S.VITS_V2_CONFIG)
    refer = S.make_refer_spec()
    xs = [torch.tensor(it["all_phones"]) for it in utt["items"]]
    berts = [it["bert"] for it in utt["items"]]
    prompts = utt["prompt_semantic"].unsqueeze(0).expand(n_utt, -1).contiguous()
    t0 = time.perf_counter()
    ys, idxs = t2s.infer_panel_batch_infer(xs, None, prompts, berts, top_k=1, top_p=1.0, temperature=1.0,
                                           early_stop_num=tokens, repetition_penalty=1.35)
    pred = [y[-i:] for y, i in zip(ys, idxs)]
    wav = vits.decode(torch.cat(pred).view(1, 1, -1), torch.cat([torch.tensor(s["phones"]) for s in segs]).view(1, -1),
                      [refer], noise=None)
    dt = time.perf_counter() - t0
    audio_s = sum(idxs) * 0.04
    return {"value": round(audio_s / dt, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
            "sample": f"{n_utt} of the batch's utterances ({sum(idxs)} tokens = {audio_s:.1f} s audio) in {dt:.1f} s, "
                      f"fp32 torch CPU oracle, AR + SoVITS decode"}


def extra_fp32(dev, B, TOK, params, segs, prompt_args, steps=3):
    """the same workload with the parity dtype (fp32 engines; the AR decode uses the launch-per-phase step)"""
    from gsv import synthetic as S
    tts = build_tts(dev, TOK, B, dtype_half=False)
    tts.set_prompt_cache(*prompt_args[0], **prompt_args[1])
    ts = []
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _sr, _a in tts.run(dict(params, segments=segs)):
            pass
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = statistics.median(ts[1:])
    audio_s = tts.last_generated_tokens * 0.04
    return {"value": round(audio_s / best, 2), "unit": "audio_s/s", "ms_per_step": round(1e3 * best, 2), "dtype": "f32",
            "steps": steps, "note": "parity dtype: token ids bit-exact vs the reference (tests/test_t2s_gpu.py)"}


def extra_v3(dev, n_steps=32, frames=934, prompt=468):
    """BASELINE configs[3]: one 934-frame chunk with a 468-frame prompt through the full DiT (1024 x 22), 32 Euler steps, then
    BigVGAN-v2 (24 kHz, x256) on the 466 new frames, fp16, random-init weights of the reference architecture."""
    from gsv import synthetic as S
    from gsv.BigVGAN.bigvgan import BigVGAN
    from gsv.f5_tts.model.backbones.dit import DiT
    from gsv.module.models import CFM
    cfg = dict(S.DIT_V3_CONFIG)
    dit = DiT(dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"], ff_mult=cfg["ff_mult"],
              mel_dim=cfg["mel_dim"], text_dim=cfg["text_dim"], conv_layers=cfg["conv_layers"], device=str(dev), dtype=torch.float16)
    dit.load_state_dict(S.make_dit_state_dict(cfg, seed=1))
    cfm = CFM(100, dit)
    voc = BigVGAN(dict(S.BIGVGAN_V2_24K_CONFIG), device=str(dev), dtype=torch.float16)
    voc.load_state_dict(S.make_vocoder_state_dict(dict(S.BIGVGAN_V2_24K_CONFIG), seed=4))
    mu = S.hash_symmetric("bench_mu", (1, frames, cfg["text_dim"]), 1.0, 1).to(dev)
    pr = S.hash_symmetric("bench_prompt", (1, 100, prompt), 1.0, 1).to(dev)
    cfm.inference(mu, None, pr, 2, seed=1)
    new = frames - prompt
    best_c, best_v = 1e9, 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel = cfm.inference(mu, None, pr, n_steps, seed=1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        wav = voc(mel[:, :, prompt:].clamp(-12, 2))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best_c, best_v = min(best_c, t1 - t0), min(best_v, t2 - t1)
    # the pipeline's batched path (TTS.using_vocoder_synthesis_batched_infer, reference TTS.py:1496-1609): the chunks of a
    # batch of segments go through ONE batched cfm.inference and are vocoded as one sequence
    NB = 8
    mu8 = mu.expand(NB, -1, -1).contiguous()
    cfm.inference(mu8, None, pr, 2, seed=1)
    best_c8, best_v8 = 1e9, 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel8 = cfm.inference(mu8, None, pr, n_steps, seed=1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m8 = mel8[:, :, prompt:].clamp(-12, 2).permute(1, 0, 2).contiguous().view(1, mel8.shape[1], -1)
        wav8 = voc(m8)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best_c8, best_v8 = min(best_c8, t1 - t0), min(best_v8, t2 - t1)
    D, inner, FF = cfg["dim"], cfg["heads"] * cfg["dim_head"], cfg["dim"] * cfg["ff_mult"]
    fl = cfg["depth"] * (2 * frames * (D * 3 * inner + inner * D + 2 * D * FF) + 4 * frames * frames * inner) \
        + 2 * 2 * frames * D * (D // 16) * 31 + 2 * frames * (2 * 100 + cfg["text_dim"]) * D + 2 * frames * D * 100
    audio_s = new * 256 / 24000.0
    return {"workload": "BASELINE configs[3]: v3 flow-matching mel (DiT 1024 x 22, 934-frame chunk, 468-frame prompt, 32 Euler "
                        "steps) + BigVGAN-v2 24 kHz x256 on the 466 new frames, fp16",
            "cfm_ms_per_euler_step": round(1e3 * best_c / n_steps, 3), "cfm_ms": round(1e3 * best_c, 2),
            "dit_tflops": round(fl * n_steps / best_c / 1e12, 1), "dit_frac_mfma": round(fl * n_steps / best_c / 2.5e15, 4),
            "vocoder_ms": round(1e3 * best_v, 2), "vocoder_tflops": round(1.80e9 * new / best_v / 1e12, 1),
            "audio_s_per_chunk": round(audio_s, 3), "value": round(audio_s / (best_c + best_v), 2), "unit": "audio_s/s",
            "finite": bool(torch.isfinite(wav).all()),
            "batched_chunks": {"chunks": NB, "cfm_ms_per_euler_step": round(1e3 * best_c8 / n_steps, 3),
                               "dit_tflops": round(NB * fl * n_steps / best_c8 / 1e12, 1),
                               "dit_frac_mfma": round(NB * fl * n_steps / best_c8 / 2.5e15, 4),
                               "vocoder_ms": round(1e3 * best_v8, 2),
                               "value": round(NB * audio_s / (best_c8 + best_v8), 2), "unit": "audio_s/s",
                               "finite": bool(torch.isfinite(wav8).all()),
                               "what": "8 chunks in one batched cfm.inference + one vocoder pass (the pipeline's parallel_infer path)"}}


def extra_cold_prompt(tts, dev):
    """reference-audio front-end before a first utterance: 8 s WAV -> 16 kHz -> HuBERT-base -> ssl_proj + VQ codes, and the
    2048-point spectrogram at 32 kHz (reference TTS.py:751-819); random-init HuBERT weights"""
    import tempfile
    import wave as _wave
    from gsv import synthetic as S
    tts.init_cnhuhbert_weights(state_dict=S.make_hubert_state_dict(seed=0))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "ref.wav")
        x = S.make_waveform(8 * 32000, 1, sr=32000).numpy()
        with _wave.open(p, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(32000)
            f.writeframes((x * 32767).astype("<i2").tobytes())
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tts.set_ref_audio(p)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return {"ms": round(1e3 * statistics.median(ts[1:]), 2), "first_call_ms": round(1e3 * ts[0], 2),
            "what": "set_ref_audio: 8 s reference (host WAV read + resample, HuBERT-base 12 x 768, extract_latent, spectrogram)"}


def extra_total_1024(sh, make_segs, B, tok_count):
    """BASELINE configs[2] at its stated size on ONE GPU: 1024 utterances in batches of B through the work queue
    (the per-GPU share of the 8-GPU job is 128; the 1 -> 8 curve itself is the driver's SCALE record)."""
    _utt, segs = make_segs(1024)
    tok_count[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sh.run(segs, batch_size=B)
    dt = time.perf_counter() - t0
    audio_s = tok_count[0] * 0.04
    return {"workload": f"BASELINE configs[2] on one GPU: 1024 utterances, {1024 // B} batches of {B} from the work queue, fp16",
            "value": round(audio_s / dt, 1), "unit": "audio_s/s", "ms": round(1e3 * dt, 1), "utterances": 1024,
            "samples": int(out.size)}


def extra_total_1024_b128(dev, TOK, params, prompt_args, make_segs):
    """the same 1024 utterances in 8 batches of 128: the AR decode of a batch is ONE persistent launch in which every row group
    serves four quads of rows (csrc/t2s_mega.hip, t2s_mega_kernel<true>); SoVITS folds the 128 utterances into one decode as
    the reference does for a batch (TTS.py:1266-1273)"""
    from gsv.sharding import ShardedSynthesizer
    B = 128
    tts = build_tts(dev, TOK, B)
    tts.set_prompt_cache(*prompt_args[0], **prompt_args[1])
    toks = [0]
    ar_ms = []

    def synth(segments):
        out = None
        for _sr, audio in tts.run(dict(params, batch_size=B, segments=segments)):
            out = audio
        toks[0] += tts.last_generated_tokens
        ar_ms.append(tts.t2s_model.decode_info())
        return out, list(tts.last_fragment_lengths)

    sh = ShardedSynthesizer(synth, dev)
    _utt, segs = make_segs(1024)
    sh.run(segs[:256], batch_size=B)                      # warm-up: kernel load, arenas, host buffers
    toks[0] = 0
    del ar_ms[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sh.run(segs, batch_size=B)
    dt = time.perf_counter() - t0
    mode, ms, steps = ar_ms[-1]
    _attn_ms, _attn_bytes, step_bytes, _layers_ms = tts.t2s_model.time_attention(iters=2)     # algorithmic bytes of one step at the final cache state
    frac = round(step_bytes / (ms / steps * 1e-3) / 8e12, 4) if mode == 1 and steps else None
    return {"workload": "BASELINE configs[2] on one GPU: 1024 utterances, 8 batches of 128 (one persistent AR launch per batch), fp16",
            "ar_step_algorithmic_bytes": int(step_bytes), "ar_step_hbm_frac": frac,
            "value": round(toks[0] * 0.04 / dt, 1), "unit": "audio_s/s", "ms": round(1e3 * dt, 1), "utterances": 1024,
            "samples": int(out.size), "ar_decode_mode": "persistent engine" if mode == 1 else "launch per phase",
            "ar_step_ms": round(ms / steps, 4) if mode == 1 and steps else None,
            "stage_ms_last_batch": {"ar_t34": round(1e3 * tts.last_timing[2], 1), "sovits_decode_t45": round(1e3 * tts.last_timing[3], 1)}}


def extra_long_form(sh, make_segs, B, tok_count, first_batch=0):
    """BASELINE configs[4] on ONE GPU: a 1400-word text = 140 sentences of 10 words, streamed in reading order (batches of B in
    submission order, `wire.streaming_generator` framing): throughput and the time to the first audible fragment."""
    import gc
    from gsv import wire
    _utt, segs = make_segs(140)
    for _ in sh.run_stream(segs[:B + max(first_batch, 1)], batch_size=B, bucket=False, first_batch=first_batch):
        pass                           # untimed: batch shapes not seen before (a first batch of 8, a last batch of 12) load their kernels
    tok_count[0] = 0
    first = [None]
    gc.collect()                       # the 1024-utterance job before this left ~10^5 host objects: a generation-2 collection inside the
    torch.cuda.synchronize()           # first batch would be charged to the time to the first fragment (seen: 57 vs 135 ms)
    t0 = time.perf_counter()

    def gen():
        for _idx, frags in sh.run_stream(segs, batch_size=B, bucket=False, first_batch=first_batch):
            if first[0] is None:
                first[0] = time.perf_counter() - t0
            yield 32000, np.concatenate(frags)

    nbytes = sum(len(c) for c in wire.streaming_generator(gen(), "wav"))
    dt = time.perf_counter() - t0
    audio_s = tok_count[0] * 0.04
    return {"workload": f"BASELINE configs[4] on one GPU: 140 sentences (1400 words) streamed in reading order, batches of {B}"
                        + (f" after a first batch of {first_batch}" if first_batch else "") + ", wav chunks",
            "value": round(audio_s / dt, 1), "unit": "audio_s/s", "ms": round(1e3 * dt, 1),
            "time_to_first_fragment_ms": round(1e3 * first[0], 1), "bytes": nbytes}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=100)
    ap.add_argument("--cpu-utts", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="parity dtype (not the benchmark configuration)")
    ap.add_argument("--total-utterances", type=int, default=0,
                    help="BASELINE configs[2]: utterances per step over the whole job, in batches of --batch (0 = --batch per GPU)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / v3 / cold-prompt sub-records")
    ap.add_argument("--pyprofile", action="store_true", help="cProfile one extra step (host-side hot spots)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gsv hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from gsv import synthetic as S
    from gsv.sharding import ShardedSynthesizer
    B, TOK = args.batch, args.tokens
    log(f"rank {rank}/{world}: building synthetic v2 checkpoints + engines")
    tts = build_tts(dev, TOK, B, dtype_half=not args.fp32)
    log("engines ready")
    total = args.total_utterances if args.total_utterances > 0 else B * world
    utt, segs_all = make_segments(total)
    prompt_args = ((utt["prompt_semantic"], [S.make_refer_spec().to(dev)]),
                   dict(phones=utt["prompt_phones"], bert_features=torch.zeros(1024, len(utt["prompt_phones"])), norm_text="x" * 40))
    tts.set_prompt_cache(*prompt_args[0], **prompt_args[1])
    params = dict(batch_size=B, top_k=1, top_p=1.0, temperature=1.0, repetition_penalty=1.35, seed=0,
                  split_bucket=True, parallel_infer=True, fragment_interval=0.3)
    tok_count = [0]

    one_stream = None
    if os.environ.get("GSV_BENCH_ONE_STREAM", "1") != "0":
        # one HIP stream for both engines and the torch glue: no cross-stream event waits between the stages
        one_stream = torch.cuda.Stream(device=dev)
        tts.t2s_model.stream = one_stream
        tts.vits_model.stream = one_stream

    def synth(segments):
        out = None
        ctx = torch.cuda.stream(one_stream) if one_stream is not None else contextlib.nullcontext()
        with ctx:
            for sr, audio in tts.run(dict(params, segments=segments)):
                out = audio
        tok_count[0] += tts.last_generated_tokens
        return out, list(tts.last_fragment_lengths)

    sh = ShardedSynthesizer(synth, dev)

    def step():
        return sh.run(segs_all if rank == 0 else None, batch_size=B if args.total_utterances > 0 else None)

    for i in range(args.warmup):
        step()
        log(f"warmup {i} done")
    if args.pyprofile:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
    # server hygiene: objects built so far (checkpoints, engines, segments) leave the collector's working set, so a
    # generation-2 collection cannot stall a timed step for ~100 ms
    import gc
    gc.collect()
    gc.freeze()
    tok_count[0] = 0
    times = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # a step is complete when step() returns: rank 0 then holds the int16 audio on the host (the D2H copy inside is
        # followed by a stream synchronize), so no device-wide synchronize per step is needed; the timed region as a whole
        # is bracketed by synchronize + barrier on both sides.  (The 20-30 ms stalls once seen here were the pageable
        # result copy, see gsv/hostcopy.py; GSV_BENCH_TRACE=1 still splits a step into run + trailing synchronize.)
        ts = time.perf_counter()
        out = step()
        t_run = time.perf_counter() - ts
        if os.environ.get("GSV_BENCH_TRACE"):
            torch.cuda.synchronize()
        times.append(time.perf_counter() - ts)
        if rank == 0 and os.environ.get("GSV_BENCH_TRACE"):
            log("   trace: sh.run %.1f ms, trailing sync %.1f ms, synth inside %.1f ms" % (1e3 * t_run, 1e3 * (times[-1] - t_run),
                                                                                      1e3 * getattr(sh, "last_synth_s", 0.0)))
        if rank == 0:
            log("step %d: %.1f ms, stages (text, to_batch, AR, SoVITS, post) = %s" % (len(times) - 1, 1e3 * times[-1],
                ", ".join("%.1f" % (1e3 * v) for v in list(tts.last_timing) + [getattr(tts, "last_postprocess_s", 0.0)])))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tot_tokens = torch.tensor([float(tok_count[0])], device=dev)
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot_tokens, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    audio_s = float(tot_tokens.item()) * 0.04            # 25 Hz semantic tokens (TTS.py:349)

    if rank == 0:
        # --- stage breakdown of the last step (reference prints the same four numbers, TTS.py:1320)
        t_text, t_batch, t_ar, t_dec = tts.last_timing
        vt_total, vt_gen = tts.vits_model.last_timing()
        # --- roofline of the kernel with the largest share of GPU time (profiles/r03_bench_kernel_stats.csv): the persistent
        # AR decode engine.  One launch covers every decode step after step 0 of the batch; its algorithmic bytes are the
        # steps' weight + K/V bytes, its duration comes from HIP events on the engine stream.
        mode, dec_ms, dec_steps = tts.t2s_model.decode_info()
        attn_ms, attn_bytes, step_bytes, layers_ms = tts.t2s_model.time_attention(iters=10)
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r03_mega_traffic.json")
        if os.path.exists(tj) and B == 32 and TOK == 100 and not args.fp32 and mode == 1:
            traffic = json.load(open(tj)).get("traffic_bytes_per_launch")
            traffic_src = "profiles/r03_mega_traffic.json (PMC passes of this command, replayed here)"
        if mode == 1 and dec_steps > 0:
            # K/V grows by one position per step: bytes at the end-of-run cache length minus the shortfall of earlier steps
            kv_pos_bytes = attn_bytes / max(1, (80 + 100 + TOK)) if attn_bytes else 0      # one cached position, all rows, one layer
            launch_bytes = dec_steps * step_bytes - 24 * kv_pos_bytes * dec_steps * (dec_steps - 1) / 2
            ach = launch_bytes / (dec_ms * 1e-3) / 1e9
            roof = {"kernel": "t2s_mega_kernel (persistent AR decode, %d steps per launch)" % dec_steps, "bound": "hbm",
                    "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(launch_bytes),
                    "avg_launch_us": round(dec_ms * 1e3, 1), "us_per_step": round(dec_ms * 1e3 / dec_steps, 1),
                    "note": "algorithmic bytes = per step 152.4 MB of weights + K/V of every cached position of every row and "
                            "layer (SURVEY 8d); HIP events on the engine stream around the one launch of the last timed step"}
        else:
            ach = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0
            roof = {"kernel": "decode_attn_kernel<f16,32> (launch-per-phase step)", "bound": "hbm", "achieved": round(ach, 1),
                    "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": int(attn_bytes), "avg_launch_us": round(attn_ms * 1e3, 2)}
        # single-utterance latency (BASELINE configs[0] shape on the GPU): median of 5 one-sentence runs
        lat = []
        for _ in range(6):
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _sr, _a in tts.run(dict(params, batch_size=1, segments=segs_all[:1])):
                pass
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - ts)
        lat_b1 = statistics.median(lat[1:])
        res = {
            "metric": "synthesised audio sec/sec (1/RTF), v2 pipeline", "value": round(audio_s / elapsed, 2),
            "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "strong" if args.total_utterances > 0 else "weak",
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: v2 t2s+SoVITS, batch=32 fixed-length sentences per GPU "
                                   "(40+40 phonemes, 100 prompt tokens, 100 generated tokens = 4.0 s each), greedy, "
                                   "random-init v2 weights", "utterances_per_gpu": B, "tokens_per_utterance": TOK,
                       "total_utterances_per_step": total, "parallelism": f"utterance-sharded x{world}"},
            "p50_utterance_latency_ms": round(1e3 * statistics.median(times), 2),
            "step_ms": [round(1e3 * t, 1) for t in times],
            "p50_single_utterance_latency_ms": round(1e3 * lat_b1, 2),
            "rtf": round(elapsed / audio_s, 6),
            "stage_ms_last_step": {"to_batch": round(1e3 * t_batch, 2), "ar_t34": round(1e3 * t_ar, 2),
                                   "sovits_decode_t45": round(1e3 * t_dec, 2), "sovits_device": round(vt_total, 2),
                                   "generator_device": round(vt_gen, 2)},
            "ar_decode_mode": "persistent engine" if mode == 1 else "launch per phase",
            "ar_step_algorithmic_bytes": int(step_bytes),
            "ar_step_ms": round(dec_ms / dec_steps, 4) if mode == 1 and dec_steps else round(layers_ms, 4),
            "ar_step_launch_path_ms": round(layers_ms, 4),
            "ar_step_hbm_frac": round(step_bytes / ((dec_ms / dec_steps if mode == 1 and dec_steps else layers_ms) * 1e-3) / 8e12, 4),
            "roofline": roof,
        }
        # --- the other stages against their bounds: one decode step, the prefill, the HiFi-GAN generator
        frames = B * TOK * 2
        gen_flop, gen_bytes = 813.2e6 * frames, 3.1e6 * frames
        res["roofline_step"] = {"bound": "hbm", "bytes": int(step_bytes), "ms": res["ar_step_ms"], "frac": res["ar_step_hbm_frac"]}
        pf_flop = 2 * 76.18e6 * B * 180 + 4 * 180 * 180 * 512 * 24 * B
        pf_ms = max(1e-3 * 0 + (1e3 * t_ar - (dec_ms if mode == 1 else layers_ms * (TOK - 1))), 1e-3)
        res["roofline_prefill"] = {"bound": "mfma", "flop": pf_flop, "ms": round(pf_ms, 2), "tflops": round(pf_flop / pf_ms / 1e9, 1),
                                   "frac": round(pf_flop / (pf_ms * 1e-3) / 2.5e15, 4),
                                   "note": "AR stage wall time minus the decode launch: prefill + step 0 + host glue"}
        res["roofline_generator"] = {"bound": "mfma+hbm", "flop": gen_flop, "ms": round(vt_gen, 2),
                                     "tflops": round(gen_flop / (vt_gen * 1e-3) / 1e12, 1),
                                     "frac_mfma": round(gen_flop / (vt_gen * 1e-3) / 2.5e15, 4),
                                     "unfused_bytes": gen_bytes, "frac_hbm": round(gen_bytes / (vt_gen * 1e-3) / 8e12, 4)}
        log(f"gpu: {res['value']} audio_s/s, {res['ms_per_step']} ms/step; roofline {roof['achieved']} GB/s")
        if not args.no_extras and world == 1 and args.total_utterances == 0 and not args.fp32:
            log("extras: configs[2] / [4] at full size on this GPU, fp32 value, v3 record, cold prompt ...")
            res["total_1024"] = extra_total_1024(sh, make_segments, B, tok_count)
            res["long_form"] = extra_long_form(sh, make_segments, B, tok_count)
            res["long_form"]["first_batch_8"] = extra_long_form(sh, make_segments, B, tok_count, first_batch=8)
            res["cold_prompt"] = extra_cold_prompt(tts, dev)
            tts = None
            sh = None
            gc.collect(); torch.cuda.empty_cache()
            res["total_1024_b128"] = extra_total_1024_b128(dev, TOK, params, prompt_args, make_segments)
            gc.collect(); torch.cuda.empty_cache()
            res["fp32"] = extra_fp32(dev, B, TOK, params, segs_all[:B], prompt_args)
            gc.collect(); torch.cuda.empty_cache()
            res["v3"] = extra_v3(dev)
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle) ...")
            res["cpu_baseline"] = cpu_baseline(args.cpu_utts, TOK)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
