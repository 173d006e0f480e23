"""CPU: host-side pipeline logic restated from the reference (TTS.to_batch bucketing, recovery_order,
audio_postprocess scaling, fragment split arithmetic, early-stop semantics) with hand-computed
expectations, plus the utterance sharding wire format (SURVEY.md section 8c last row, section 8e)."""
import numpy as np
import pytest
import torch

from gsv.TTS_infer_pack.TTS import TTS, TTS_Config
from gsv import sharding


def _tts_cpu():
    """host logic only: construct the pipeline object without engines"""
    cfg = TTS_Config({"device": "cpu", "is_half": False, "version": "v2"})
    return TTS(cfg)


def _seg(n, text_len=None):
    return {"phones": list(range(n)), "bert_features": torch.zeros(1024, n), "norm_text": "x" * (text_len or n)}


def test_to_batch_bucketing_hand_computed():
    tts = _tts_cpu()
    # text lengths 10,2,9,30,3,11 ; batch_size 3 ; threshold 0.75
    data = [_seg(4, L) for L in (10, 2, 9, 30, 3, 11)]
    batches, index = tts.to_batch(data, None, batch_size=3, threshold=0.75, split_bucket=True)
    # sorted by length: idx [1(2),4(3),2(9),0(10),5(11),3(30)]
    # pos0: cand [2,3,9] median(elem 1)=3 mean=4.67 -> 0.64 < .75 shrink -> [2,3] elem1=3 mean 2.5 -> 1.2 ok
    # pos2: cand [9,10,11] elem1=10 mean=10 -> ok ; pos5: [30] single
    assert index == [[1, 4], [2, 0, 5], [3]]
    assert [len(b["phones"]) for b in batches] == [2, 3, 1]
    assert sum(len(b) for b in index) == len(data)
    # no bucketing: submission order in chunks
    _, index2 = tts.to_batch(data, None, batch_size=4, split_bucket=False)
    assert index2 == [[0, 1, 2, 3], [4, 5]]


def test_to_batch_prepends_prompt_and_reports_max_len():
    tts = _tts_cpu()
    prompt = {"phones": [7, 8, 9], "bert_features": torch.ones(1024, 3)}
    data = [_seg(2), _seg(5)]
    batches, index = tts.to_batch(data, prompt, batch_size=5, threshold=0.0)
    b = batches[0]
    assert [t.tolist() for t in b["all_phones"]] == [[7, 8, 9, 0, 1], [7, 8, 9, 0, 1, 2, 3, 4]]
    assert b["all_phones_len"].tolist() == [5, 8] and b["phones_len"].tolist() == [2, 5]
    assert b["all_bert_features"][1].shape == (1024, 8) and b["max_len"] == 8
    assert float(b["all_bert_features"][0][:, :3].min()) == 1.0 and float(b["all_bert_features"][0][:, 3:].max()) == 0.0


def test_recovery_order_and_postprocess_scaling():
    tts = _tts_cpu()
    index = [[2, 0], [1]]
    audio = [[torch.full((4,), 0.5), torch.tensor([2.0, -4.0, 1.0])], [torch.tensor([0.25, -0.25])]]
    sr, out = tts.audio_postprocess(audio, 32000, index, 1.0, True, fragment_interval=0.0001)  # 3 zeros
    assert sr == 32000 and out.dtype == np.int16
    # order restored: item0 = second fragment of batch 0 (peak 4 > 1 -> divided by 4), item1, item2
    exp = [0.5, -1.0, 0.25, 0, 0, 0, 0.25, -0.25, 0, 0, 0, 0.5, 0.5, 0.5, 0.5, 0, 0, 0]
    assert out.tolist() == (np.array(exp) * 32768).astype(np.int16).tolist()
    assert tts.last_fragment_lengths == [6, 5, 7]
    # -1.0 * 32768 wraps/clamps exactly like numpy astype on the reference (TTS.py:1421)
    assert out[1] == np.array([-32768.0]).astype(np.int16)[0]


def test_deal_contiguous_and_wire_format_roundtrip():
    lens = [5, 1, 9, 3, 7, 2, 8]
    shares = sharding.deal_contiguous(lens, 3)
    assert sorted(sum(shares, [])) == list(range(7))
    assert [len(s) for s in shares] == [3, 2, 2]
    assert [lens[i] for i in shares[0]] == [1, 2, 3]          # shortest run on rank 0
    segs = [_seg(n, n + 1) for n in (3, 1, 4)]
    back = sharding.unpack_segments(sharding.pack_segments(segs))
    assert [b["phones"] for b in back] == [s["phones"] for s in segs]
    assert [len(b["norm_text"]) for b in back] == [4, 2, 5]
    assert sharding.unpack_segments(sharding.pack_segments([])) == []


def _stub_synth(segments):
    """fake engine: fragment i = int16 ramp of length 10*len(phones) filled with phones[0]"""
    frags = [torch.full((10 * len(s["phones"]),), s["phones"][0] if s["phones"] else 0, dtype=torch.int16) for s in segments]
    return (torch.cat(frags) if frags else torch.zeros(0, dtype=torch.int16)), [int(f.numel()) for f in frags]


def test_sharded_synthesizer_single_process_restores_order():
    segs = [{"phones": [i + 1] * n, "bert_features": torch.zeros(1024, n), "norm_text": "x" * n} for i, n in enumerate((4, 1, 3))]
    out = sharding.ShardedSynthesizer(_stub_synth, torch.device("cpu")).run(segs)
    assert out.tolist() == [1] * 40 + [2] * 10 + [3] * 30


def _gloo_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        segs = None
        if rank == 0:
            segs = [{"phones": [i + 1] * n, "bert_features": torch.zeros(1024, n), "norm_text": "x" * n}
                    for i, n in enumerate((4, 1, 3, 6, 2))]
        out = sharding.ShardedSynthesizer(_stub_synth, torch.device("cpu")).run(segs)
        q.put((rank, None if out is None else out.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_synthesizer_gloo_world(world):
    """N > 1 path on CPU: scatter (broadcast + slice), local synthesis, padded gather, order restore."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world * 7 + (os_getpid() % 200)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = [1] * 40 + [2] * 10 + [3] * 30 + [4] * 60 + [5] * 20
    assert res[0] == exp
    assert all(res[r] is None for r in range(1, world))


def os_getpid():
    import os
    return os.getpid()


def test_run_error_protocol_without_engines():
    """reference TTS.py:1352-1363: on failure yield 1 s of silence at 16 kHz, then re-raise"""
    tts = _tts_cpu()
    gen = tts.run({"segments": [_seg(3)]})
    sr, audio = next(gen)
    assert sr == 16000 and audio.shape == (16000,) and audio.dtype == np.int16 and not audio.any()
    with pytest.raises(RuntimeError):
        next(gen)


def test_text_segmentation_methods_match_reference_outputs():
    """N1: cut0..cut5 / split / split_big_text against outputs of the reference module
    (TTS_infer_pack/text_segmentation_method.py, run in the build container; tests/golden/text_segmentation.json)."""
    import json
    import os
    from gsv.TTS_infer_pack import text_segmentation_method as seg
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_segmentation.json"), encoding="utf-8"))
    texts = g["texts"]
    assert sorted(seg.get_method_names()) == sorted(g["methods"])
    for name, outs in g["methods"].items():
        fn = seg.get_method(name)
        for t, want in zip(texts, outs):
            assert fn(t) == want, (name, t)
    for t, want in zip([t for t in texts if t.strip("\n")], g["split"]):
        assert seg.split(t) == want
    for t, want in zip(texts, g["split_big_text"]):
        assert seg.split_big_text(t, 40) == want
    with pytest.raises(ValueError):
        seg.get_method("cut9")
