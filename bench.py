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

Prints ONE JSON line (rank 0).  `roofline` = the AR decode-attention kernel (HBM-bound, SURVEY.md
section 8d): algorithmic KV bytes per launch / average launch time measured with HIP events inside the
library on the engine's stream (gsv_t2s_time_step).  `cpu_baseline` = the oracle (CPU restatement,
kind "port") on a bounded sample of the same workload on this box's host cores.
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
    utt, segs_all = make_segments(B * world)
    tts.set_prompt_cache(utt["prompt_semantic"], [S.make_refer_spec().to(dev)], phones=utt["prompt_phones"],
                         bert_features=torch.zeros(1024, len(utt["prompt_phones"])), norm_text="x" * 40)
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
        return sh.run(segs_all if rank == 0 else None)

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
        # --- roofline of the dominant AR kernel: decode attention streaming the KV arena
        attn_ms, attn_bytes, step_bytes, layers_ms = tts.t2s_model.time_attention(iters=10)
        ach = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE collected
        # in separate rocprofv3 runs of this command, gfx950 2x read correction applied); valid for this workload
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_attn_traffic.json")
        if os.path.exists(tj) and B == 32 and TOK == 100 and not args.fp32:
            traffic = json.load(open(tj)).get("traffic_bytes_per_launch")
        roof = {"kernel": "decode_attn_kernel<f16,32>", "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(attn_bytes), "avg_launch_us": round(attn_ms * 1e3, 2),
                "note": "one launch = one layer's K+V arena for all rows at the end-of-run cache length; "
                        "HIP events on the engine stream around 10x24 back-to-back launches over the 24 layers' "
                        "distinct arenas (24 x 18 MB > Infinity Cache: every launch streams from HBM), rows forced active"}
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
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: v2 t2s+SoVITS, batch=32 fixed-length sentences per GPU "
                                   "(40+40 phonemes, 100 prompt tokens, 100 generated tokens = 4.0 s each), greedy, "
                                   "random-init v2 weights", "utterances_per_gpu": B, "tokens_per_utterance": TOK,
                       "parallelism": f"utterance-sharded x{world}"},
            "p50_utterance_latency_ms": round(1e3 * statistics.median(times), 2),
            "step_ms": [round(1e3 * t, 1) for t in times],
            "p50_single_utterance_latency_ms": round(1e3 * lat_b1, 2),
            "rtf": round(elapsed / audio_s, 6),
            "stage_ms_last_step": {"to_batch": round(1e3 * t_batch, 2), "ar_t34": round(1e3 * t_ar, 2),
                                   "sovits_decode_t45": round(1e3 * t_dec, 2), "sovits_device": round(vt_total, 2),
                                   "generator_device": round(vt_gen, 2)},
            "ar_step_algorithmic_bytes": int(step_bytes), "ar_step_layers_eager_ms": round(layers_ms, 4),
            "ar_step_hbm_frac": round(step_bytes / (layers_ms * 1e-3) / 8e12, 4) if layers_ms > 0 else None,
            "roofline": roof,
        }
        log(f"gpu: {res['value']} audio_s/s, {res['ms_per_step']} ms/step; roofline {roof['achieved']} GB/s")
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle) ...")
            res["cpu_baseline"] = cpu_baseline(args.cpu_utts, TOK)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
