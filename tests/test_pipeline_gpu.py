"""GPU: the whole v2 path through TTS.run (to_batch -> AR -> time-folded SoVITS decode -> post-process)
against the same chain built from the oracles, on reduced models in fp32."""
import math

import numpy as np
import pytest
import torch

from oracle import cases
from oracle.t2s_oracle import T2SOracle
from oracle.vits_oracle import VitsOracle
from gsv import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(is_half=False, max_sec=0.4):
    from gsv.TTS_infer_pack.TTS import TTS
    tcfg = S.small_t2s_config(n_layer=2, dim=128, head=4, vocab=1025, phoneme_vocab=732)
    tcfg["data"]["max_sec"] = max_sec                       # early_stop_num = 50 * max_sec = 20 tokens
    tsd = S.make_t2s_state_dict(tcfg, seed=11, suppress_eos=True)
    vcfg = S.small_vits_config()
    vsd = S.make_vits_state_dict(vcfg, seed=12)
    tts = TTS({"device": DEV, "is_half": is_half, "version": "v2", "max_batch": 4, "max_seq": 256})
    tts.init_t2s_weights(state={"weight": tsd, "config": tcfg})
    tts.init_vits_weights(state={"weight": vsd, "config": vcfg})
    return tts, tcfg, tsd, vcfg, vsd


def test_tts_run_matches_oracle_chain():
    tts, tcfg, tsd, vcfg, vsd = _build()
    utt = S.make_utterances(5, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=21, ragged=True)
    for i, it in enumerate(utt["items"]):                  # distinct text lengths -> bucketing reorders
        it["norm_text"] = "x" * [7, 3, 9, 4, 8][i]
    refer = S.make_refer_spec(frames=30, seed=5)
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": it["norm_text"]}
            for it in utt["items"]]
    tts.set_prompt_cache(utt["prompt_semantic"], [refer.to(DEV)], phones=utt["prompt_phones"],
                         bert_features=torch.zeros(1024, 6), norm_text="xxxxxx")
    # inject the SoVITS noise (TTS.run itself uses the counter RNG): one draw per decode call
    real_decode = tts.vits_model.decode
    IC = vcfg["model"]["inter_channels"]
    noises = []

    def decode_with_noise(codes, text, refer_, **kw):
        nz = S.hash_normal(f"pipe_noise{len(noises)}", (IC, 2 * codes.shape[-1]), 0)
        noises.append(nz)
        kw.pop("seed", None)
        return real_decode(codes, text, refer_, noise=nz, **kw)

    tts.vits_model.decode = decode_with_noise
    out = list(tts.run({"segments": segs, "batch_size": 2, "top_k": 1, "top_p": 1.0, "temperature": 1.0,
                        "repetition_penalty": 1.35, "seed": 3, "split_bucket": True, "fragment_interval": 0.01}))
    assert len(out) == 1
    sr, audio = out[0]
    assert sr == 32000 and audio.dtype == np.int16

    # ---- the same chain from the oracles
    t2s, vits = T2SOracle(tsd, tcfg), VitsOracle(vsd, vcfg)
    data, index = tts.to_batch(segs, {"phones": utt["prompt_phones"], "bert_features": torch.zeros(1024, 6)},
                               batch_size=2, threshold=0.75, split_bucket=True)
    up = math.prod(vcfg["model"]["upsample_rates"])
    frags_by_batch = []
    for bi, item in enumerate(data):
        n = len(item["all_phones"])
        berts = [torch.zeros(1024, int(t.shape[0])) for t in item["all_phones"]]
        ys, idxs = t2s.infer_panel_batch_infer(item["all_phones"], None, utt["prompt_semantic"].view(1, -1).expand(n, -1),
                                               berts, top_k=1, top_p=1.0, temperature=1.0, early_stop_num=20,
                                               repetition_penalty=1.35)
        pred = [y[-i:] for y, i in zip(ys, idxs)]
        wav = vits.decode(torch.cat(pred).view(1, 1, -1), torch.cat(item["phones"]).view(1, -1), [refer], noise=noises[bi])[0, 0]
        fr, o = [], 0
        for p in pred:
            fr.append(wav[o:o + p.shape[0] * 2 * up])
            o += p.shape[0] * 2 * up
        frags_by_batch.append(fr)
    zero = torch.zeros(int(32000 * 0.01))
    post = [[torch.cat([f / max(1.0, float(f.abs().max())), zero]) for f in fr] for fr in frags_by_batch]
    flat = tts.recovery_order(post, index)
    ref = (torch.cat(flat).numpy() * 32768).astype(np.int16)
    assert audio.shape == ref.shape
    assert np.abs(audio.astype(np.int32) - ref.astype(np.int32)).max() <= 4      # fp32 engine: waveform <= 1e-4 -> <= 4 LSB
    assert tts.last_generated_tokens == 5 * 20


def test_tts_run_error_protocol_and_stop_on_gpu():
    tts, *_ = _build()
    # no prompt cache -> NO_PROMPT_ERROR after one second of silence (reference TTS.py:1078, 1352-1363)
    from gsv.TTS_infer_pack.TTS import NO_PROMPT_ERROR
    gen = tts.run({"segments": [{"phones": [1, 2, 3], "bert_features": None, "norm_text": "abc"}]})
    sr, a = next(gen)
    assert sr == 16000 and a.shape == (16000,) and not a.any()
    with pytest.raises(NO_PROMPT_ERROR):
        next(gen)
    # engines were rebuilt by the error protocol and still work; empty segment list -> 1 s of silence
    assert tts.t2s_model is not None and tts.vits_model is not None
    sr, a = next(tts.run({"segments": []}))
    assert sr == 16000 and not a.any()


def test_get_tts_wav_cli_path_matches_oracle_tokens_and_scaling():
    """inference_cli / get_tts_wav glue (naive AR entry point, per-sentence decode, x32767)."""
    from gsv.inference_cli import get_tts_wav
    tts, tcfg, tsd, vcfg, vsd = _build(max_sec=0.3)
    utt = S.make_utterances(2, prompt_phones=5, target_phones=8, prompt_tokens=7, seed=31)
    refer = S.make_refer_spec(frames=25, seed=6)
    prompt_seg = {"phones": utt["prompt_phones"], "bert_features": None}
    segs = [{"phones": it["phones"], "bert_features": None, "norm_text": it["norm_text"]} for it in utt["items"]]
    res = list(get_tts_wav(tts.t2s_model, tts.vits_model, utt["prompt_semantic"], refer.to(DEV), prompt_seg, segs,
                           top_k=1, max_sec=0.3, hz=50))
    assert len(res) == 1
    sr, audio = res[0]
    # token count: naive loop masks EOS for 11 steps; early stop at 15 tokens per sentence
    t2s = T2SOracle(tsd, tcfg)
    n_tok = 0
    for it in utt["items"]:
        ids = torch.LongTensor(utt["prompt_phones"] + it["phones"]).unsqueeze(0)
        y, idx = t2s.infer_panel_naive(ids, None, utt["prompt_semantic"].view(1, -1), torch.zeros(1, 1024, ids.shape[1]),
                                       top_k=1, top_p=1.0, temperature=1.0, early_stop_num=15)
        n_tok += idx
    up = math.prod(vcfg["model"]["upsample_rates"])
    assert audio.shape[0] == n_tok * 2 * up + 2 * int(32000 * 0.3)
    assert sr == 32000 and audio.dtype == np.int16 and np.abs(audio).max() <= 32767


def test_tts_run_prompt_free_matches_oracle_chain():
    """no prompt text (reference TTS.py:1124-1131): nothing is prepended, the AR decoder starts from an empty audio prefix
    (naive loop per sentence, EOS masked for 11 steps) and the whole generated sequence goes to the SoVITS decoder."""
    tts, tcfg, tsd, vcfg, vsd = _build()
    utt = S.make_utterances(3, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=23, ragged=True)
    refer = S.make_refer_spec(frames=30, seed=5)
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": "x" * (5 + i)}
            for i, it in enumerate(utt["items"])]
    tts.set_prompt_cache(None, [refer.to(DEV)], phones=None)
    real_decode = tts.vits_model.decode
    IC = vcfg["model"]["inter_channels"]
    noises = []

    def decode_with_noise(codes, text, refer_, **kw):
        nz = S.hash_normal(f"pf_noise{len(noises)}", (IC, 2 * codes.shape[-1]), 0)
        noises.append(nz)
        kw.pop("seed", None)
        return real_decode(codes, text, refer_, noise=nz, **kw)

    tts.vits_model.decode = decode_with_noise
    out = list(tts.run({"segments": segs, "batch_size": 3, "top_k": 1, "seed": 3, "split_bucket": False, "fragment_interval": 0.01}))
    sr, audio = out[0]
    t2s, vits = T2SOracle(tsd, tcfg), VitsOracle(vsd, vcfg)
    up = math.prod(vcfg["model"]["upsample_rates"])
    pred = []
    for it in segs:
        ph = torch.tensor(it["phones"])
        y, idx = t2s.infer_panel_naive(ph.unsqueeze(0), None, None, torch.zeros(1, 1024, ph.shape[0]), top_k=1, top_p=1.0,
                                       temperature=1.0, early_stop_num=20, repetition_penalty=1.35)
        assert idx == 0
        pred.append(y[0])
    wav = vits.decode(torch.cat(pred).view(1, 1, -1), torch.cat([torch.tensor(it["phones"]) for it in segs]).view(1, -1), [refer],
                      noise=noises[0])[0, 0]
    zero = torch.zeros(int(32000 * 0.01))
    parts, o = [], 0
    for p in pred:
        f = wav[o:o + p.shape[0] * 2 * up]
        o += p.shape[0] * 2 * up
        parts += [f / max(1.0, float(f.abs().max())), zero]
    ref = (torch.cat(parts) * 32768).to(torch.int32).to(torch.int16).numpy()
    assert audio.shape == ref.shape
    assert np.abs(audio.astype(np.int32) - ref.astype(np.int32)).max() <= 4


def test_none_bert_features_equal_zero_features():
    """the sharded path's wire format carries no BERT block: `bert_features=None` on a segment must give exactly the
    audio of the all-zero [1024, X] tensor the reference's non-zh front-end produces (TextPreprocessor.py:216-220)."""
    tts, *_ = _build()
    utt = S.make_utterances(3, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=23, ragged=True)
    refer = S.make_refer_spec(frames=30, seed=5)
    tts.set_prompt_cache(utt["prompt_semantic"], [refer.to(DEV)], phones=utt["prompt_phones"],
                         bert_features=torch.zeros(1024, 6), norm_text="xxxxxx")
    params = {"batch_size": 2, "top_k": 1, "top_p": 1.0, "temperature": 1.0, "repetition_penalty": 1.35, "seed": 3,
              "split_bucket": True, "fragment_interval": 0.01}
    outs = []
    for zero in (True, False):
        segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])) if zero else None,
                 "norm_text": it["norm_text"]} for it in utt["items"]]
        (sr, audio), = list(tts.run(dict(params, segments=segs)))
        outs.append(audio.copy())
    assert outs[0].size > 0 and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("half", [False, True])
def test_audio_postprocess_kernel_matches_reference_arithmetic(half):
    """H13 (`gsv_postprocess`) against the reference's host arithmetic (TTS.py:1377-1429) in the fragments' own dtype:
    peak division only above 1, silence gaps, order recovery, x32768 in that dtype, numpy's astype(int16) wrap;
    bit-exact, including 70 fragments (three table launches), an empty fragment and a fragment holding a NaN."""
    from gsv.TTS_infer_pack.TTS import TTS
    tts = TTS({"device": DEV, "is_half": half, "version": "v2"})
    dt, npdt = (torch.float16, np.float16) if half else (torch.float32, np.float32)
    g = torch.Generator().manual_seed(5)
    # hand-computed case first (the old host-logic test): item0 has peak 4 -> divided, -1.0 * 32768 wraps like numpy
    index = [[2, 0], [1]]
    audio = [[torch.full((4,), 0.5), torch.tensor([2.0, -4.0, 1.0])], [torch.tensor([0.25, -0.25])]]
    sr, out = tts.audio_postprocess([[f.to(DEV, dt) for f in b] for b in audio], 32000, index, 1.0, True, fragment_interval=0.0001)
    exp = [0.5, -1.0, 0.25, 0, 0, 0, 0.25, -0.25, 0, 0, 0, 0.5, 0.5, 0.5, 0.5, 0, 0, 0]
    assert sr == 32000 and out.dtype == np.int16 and out.tolist() == (np.array(exp) * 32768).astype(np.int16).tolist()
    assert tts.last_fragment_lengths == [6, 5, 7] and out[1] == -32768
    # many ragged fragments in shuffled batches
    lens = [int(v) for v in torch.randint(1, 5000, (70,), generator=g)]
    lens[7] = 0
    frags = [((torch.rand(n, generator=g) * 2 - 1) * (0.3 if i % 3 else 1.7)).to(dt) for i, n in enumerate(lens)]
    frags[11][3] = float("nan")
    perm = torch.randperm(70, generator=g).tolist()
    index = [perm[:32], perm[32:50], perm[50:]]
    batches = [[frags[i].to(DEV) for i in ix] for ix in index]
    gap = int(32000 * 0.01)
    sr, out = tts.audio_postprocess(batches, 32000, index, 1.0, True, fragment_interval=0.01)
    parts = []
    for f in frags:                                             # reference loop, numpy in the fragment dtype
        a = f.numpy().astype(npdt)
        m = np.abs(a).max() if a.size else 0
        if m > 1:
            a = a / m
        parts += [a, np.zeros(gap, npdt)]
    ref = np.concatenate(parts)
    with np.errstate(invalid="ignore"):
        ref16 = (ref * 32768).astype(np.int32).astype(np.int16)
    nan_at = np.isnan(ref)
    assert out.shape == ref16.shape and np.array_equal(out[~nan_at], ref16[~nan_at])
    assert tts.last_fragment_lengths == [n + gap for n in lens]


def test_sharded_stream_of_a_long_job_on_one_gpu():
    """BASELINE configs[2] / [4] code path on one GPU (world size 1; the multi-rank exchange is covered by the gloo tests):
    11 segments -> length-sorted batches of 4 through the work queue -> fragments come back in submission order, equal to a
    direct TTS.run of each batch, and `wire.streaming_generator` frames them as one wav header + raw chunks."""
    from gsv import sharding, wire
    tts, tcfg, tsd, vcfg, vsd = _build()
    utt = S.make_utterances(11, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=5, ragged=True)
    refer = S.make_refer_spec(frames=30, seed=5)
    tts.set_prompt_cache(utt["prompt_semantic"], [refer.to(DEV)], phones=utt["prompt_phones"], bert_features=torch.zeros(1024, 6),
                         norm_text="xxxxxx")
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": "x" * (3 + (7 * i) % 5)}
            for i, it in enumerate(utt["items"])]
    params = dict(batch_size=4, top_k=1, seed=0, split_bucket=True, parallel_infer=True, fragment_interval=0.01)

    def synth(batch):
        out = None
        for sr, audio in tts.run(dict(params, segments=batch)):
            out = audio
        return out, list(tts.last_fragment_lengths)

    sh = sharding.ShardedSynthesizer(synth, torch.device(DEV))
    got = {}
    order = []
    for idxs, frags in sh.run_stream(segs, batch_size=4):
        order.append(list(idxs))
        for i, f in zip(idxs, frags):
            got[i] = f
    batches = sharding.make_batches([len(s["norm_text"]) for s in segs], 4)
    assert order == batches and sorted(got) == list(range(11))
    for b in batches:                                           # every batch equals a direct run of the same segments
        ref, lens = synth([segs[i] for i in b])
        o = 0
        for i, n in zip(b, lens):
            assert np.array_equal(got[i], ref[o:o + n])
            o += n
    whole = sh.run(segs, batch_size=4)
    assert whole.dtype == np.int16 and whole.size == sum(len(got[i]) for i in range(11))
    assert np.array_equal(whole, np.concatenate([got[i] for i in range(11)]))
    chunks = list(wire.streaming_generator(((32000, np.concatenate(fr)) for _, fr in sh.run_stream(segs, batch_size=4)), "wav"))
    assert chunks[0][:4] == b"RIFF" and len(chunks[0]) == 44 and len(chunks) == 1 + len(batches)
    assert b"".join(chunks[1:]) == np.concatenate([got[i] for b in batches for i in b]).tobytes()


def test_return_fragment_streams_batches_and_http_seam():
    """`return_fragment` (TTS.py:1321-1330): one yield per batch in submission order (bucketing off, as the reference forces), whose
    concatenation equals the one-shot result with split_bucket=False; and `wire.tts_handle` in streaming mode on the real pipeline:
    44-byte wav header chunk, then the same samples as raw chunks; a failing request behaves as in the reference."""
    from gsv import wire
    tts, *_ = _build()
    utt = S.make_utterances(5, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=9, ragged=True)
    refer = S.make_refer_spec(frames=30, seed=5)
    tts.set_prompt_cache(utt["prompt_semantic"], [refer.to(DEV)], phones=utt["prompt_phones"], bert_features=torch.zeros(1024, 6),
                         norm_text="xxxxxx")
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": "x" * (4 + i)}
            for i, it in enumerate(utt["items"])]
    req = dict(segments=segs, batch_size=2, top_k=1, seed=0, fragment_interval=0.01, parallel_infer=True)
    parts = list(tts.run(dict(req, return_fragment=True)))
    assert len(parts) == 3 and all(sr == 32000 and a.dtype == np.int16 for sr, a in parts)
    sr, whole = list(tts.run(dict(req, split_bucket=False)))[-1]
    assert np.array_equal(np.concatenate([a for _, a in parts]), whole)
    code, mt, it = wire.tts_handle(tts, dict(req, streaming_mode=True, media_type="wav"))
    chunks = list(it)
    assert code == 200 and mt == "audio/wav" and len(chunks) == 4 and len(chunks[0]) == 44 and chunks[0][:4] == b"RIFF"
    assert b"".join(chunks[1:]) == whole.tobytes()
    code, mt, body = wire.tts_handle(tts, dict(req, media_type="raw", split_bucket=False))
    assert code == 200 and body == whole.tobytes()
    tts.prompt_cache["prompt_semantic"] = None
    tts.prompt_cache["refer_spec"] = []
    # a failing request: TTS.run yields one second of silence and raises on the NEXT item (TTS.py:1352-1363); the reference's
    # non-streaming handler only takes the first item (api_v2.py:356-358), so it answers 200 with that silence -- mirrored
    code, mt, body = wire.tts_handle(tts, dict(req, media_type="raw"))
    assert code == 200 and body == bytes(32000)
    code, mt, it = wire.tts_handle(tts, dict(req, streaming_mode=True, media_type="raw"))
    with pytest.raises(Exception):
        list(it)
