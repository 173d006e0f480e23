"""CPU: host-side pipeline logic restated from the reference (TTS.to_batch bucketing, recovery_order,
audio_postprocess scaling, fragment split arithmetic, early-stop semantics) with hand-computed
expectations, plus the utterance sharding wire format (SURVEY.md section 8c last row, section 8e)."""
import os
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


def test_recovery_order():
    """TTS.py:957-982: fragments return to the input order (the peak / int16 stage is a HIP kernel: tests/test_pipeline_gpu.py)"""
    tts = _tts_cpu()
    index = [[2, 0], [1]]
    audio = [[torch.full((4,), 0.5), torch.tensor([2.0, -4.0, 1.0])], [torch.tensor([0.25, -0.25])]]
    flat = tts.recovery_order(audio, index)
    assert [f.tolist() for f in flat] == [[2.0, -4.0, 1.0], [0.25, -0.25], [0.5] * 4]
    with pytest.raises(Exception):                       # no CPU fallback for the post-processing kernel
        tts.audio_postprocess(audio, 32000, index, 1.0, True, fragment_interval=0.0001)


def test_to_batch_and_recovery_order_match_the_reference_methods():
    """TTS.to_batch / TTS.recovery_order against fixtures written by the REFERENCE's own methods (TTS.py:842-973, called
    unbound by oracle/gen_golden_tts_glue.py): bucket composition and order, per-batch max_len, prompt-prefixed phones."""
    from conftest import load_golden
    from oracle import glue_cases as G
    g = load_golden("tts_glue_host")
    tts = _tts_cpu()
    for ci, (lens, bs, thr, sb) in enumerate(G.TO_BATCH_CASES):
        data, prompt_data = G.to_batch_data(lens)
        batches, index = tts.to_batch(data, prompt_data, batch_size=bs, threshold=thr, split_bucket=sb)
        assert [i for b in index for i in b] == g[f"tb{ci}_index"].tolist(), f"case {ci}: order"
        assert [len(b) for b in index] == g[f"tb{ci}_sizes"].tolist(), f"case {ci}: bucket sizes"
        assert [int(b["max_len"]) for b in batches] == g[f"tb{ci}_max_len"].tolist()
        assert torch.cat([b["all_phones_len"] for b in batches]).tolist() == g[f"tb{ci}_all_len"].tolist()
        assert torch.cat([p for b in batches for p in b["all_phones"]]).tolist() == g[f"tb{ci}_all_phones"].tolist()
        assert tts.recovery_order([[str(i) for i in b] for b in index], index) == [str(i) for i in range(len(lens))]


def test_make_batches_first_batch_and_reading_order():
    lens = [5, 1, 4, 2, 3, 9, 8]
    assert sharding.make_batches(lens, 3) == [[1, 3, 4], [2, 0, 6], [5]]
    assert sharding.make_batches(lens, 3, bucket=False) == [[0, 1, 2], [3, 4, 5], [6]]
    assert sharding.make_batches(lens, 3, bucket=False, first_batch=1) == [[0], [1, 2, 3], [4, 5, 6]]
    assert sharding.make_batches(lens, 3, bucket=False, first_batch=3) == [[0, 1, 2], [3, 4, 5], [6]]      # not smaller: ignored
    got = list(sharding.ShardedSynthesizer(_stub_synth, torch.device("cpu")).run_stream(
        [{"phones": [i + 1] * 2, "bert_features": None, "norm_text": "x" * n} for i, n in enumerate(lens)], batch_size=3,
        bucket=False, first_batch=2))
    assert [idx for idx, _ in got] == [[0, 1], [2, 3, 4], [5, 6]]


def test_deal_contiguous_batches_and_wire_format_roundtrip():
    lens = [5, 1, 9, 3, 7, 2, 8]
    shares = sharding.deal_contiguous(lens, 3)
    assert sorted(sum(shares, [])) == list(range(7))
    assert [len(s) for s in shares] == [3, 2, 2]
    assert [lens[i] for i in shares[0]] == [1, 2, 3]          # shortest run on rank 0
    assert sharding.make_batches(lens, 3) == [[1, 5, 3], [0, 4, 6], [2]]      # length-sorted runs of 3
    # BASELINE configs[2]: 1024 utterances -> 32 batches of 32
    assert [len(b) for b in sharding.make_batches([40] * 1024, 32)] == [32] * 32
    segs = [_seg(n, n + 1) for n in (3, 1, 4)]
    wire, bert = sharding.pack_segments(segs)
    assert bert is None                                         # all-zero BERT features are never shipped
    back = sharding.unpack_segments(wire, bert)
    assert [b["phones"] for b in back] == [s["phones"] for s in segs]
    assert [len(b["norm_text"]) for b in back] == [4, 2, 5] and all(b["bert_features"] is None for b in back)
    assert sharding.unpack_segments(*sharding.pack_segments([])) == []
    # zh segments: non-zero features travel (fp16) -- or are refused loudly, never dropped
    segs[1]["bert_features"] = torch.arange(1024, dtype=torch.float32).unsqueeze(1) / 1024
    wire, bert = sharding.pack_segments(segs)
    assert tuple(bert.shape) == (1024, 1)
    back = sharding.unpack_segments(wire, bert)
    assert back[0]["bert_features"] is None and torch.allclose(back[1]["bert_features"], segs[1]["bert_features"], atol=1e-3)
    with pytest.raises(ValueError):
        sharding.pack_segments(segs, ship_bert=False)


def _stub_synth(segments):
    """fake engine: fragment i = int16 ramp of length 10*len(phones) filled with phones[0] (+ 100 when BERT features arrived)"""
    frags = [torch.full((10 * len(s["phones"]),), (s["phones"][0] if s["phones"] else 0) + (100 if s["bert_features"] is not None and bool(s["bert_features"].any()) else 0),
                        dtype=torch.int16) for s in segments]
    return (torch.cat(frags) if frags else torch.zeros(0, dtype=torch.int16)), [int(f.numel()) for f in frags]


def test_join_fragments_is_a_view_only_when_fragments_are_adjacent_in_order():
    """the one-batch job of BASELINE configs[1] returns the page-locked result buffer itself; every other layout is copied"""
    a = np.arange(100, dtype=np.int16)
    v = sharding.join_fragments([a[0:10], a[10:10], a[10:30]])
    assert np.shares_memory(v, a) and v.tolist() == list(range(30))
    for frags in ([a[10:30], a[0:10]], [a[0:10], a[11:30]], [a[0:10], a.copy()[10:30]], [a.copy(), a.copy()], [a[0:10:2]]):
        v = sharding.join_fragments(frags)
        assert not np.shares_memory(v, a) and v.tolist() == np.concatenate(frags).tolist()
    m = np.arange(40, dtype=np.int16).reshape(4, 10)                     # rows of a 2-D buffer: not the 1-D layout, copied
    assert sharding.join_fragments([m[0], m[1]]).tolist() == list(range(20))
    assert sharding.join_fragments([]).size == 0 and sharding.join_fragments([a[3:3], a[5:5]]).size == 0
    bufs = []

    def synth(segs):
        bufs.append(np.arange(1000 * len(segs), dtype=np.int16))
        return bufs[-1], [1000] * len(segs)
    sh = sharding.ShardedSynthesizer(synth, torch.device("cpu"))
    segs = [_seg(5) for _ in range(8)]
    out = sh.run(segs)
    assert np.shares_memory(out, bufs[0]) and out.tolist() == bufs[0].tolist()
    out = sh.run(segs, batch_size=4)                                     # two batches = two buffers: concatenated
    assert out.tolist() == bufs[1].tolist() + bufs[2].tolist()


def test_sharded_synthesizer_single_process_restores_order():
    segs = [{"phones": [i + 1] * n, "bert_features": torch.zeros(1024, n), "norm_text": "x" * n} for i, n in enumerate((4, 1, 3))]
    sh = sharding.ShardedSynthesizer(_stub_synth, torch.device("cpu"))
    assert sh.run(segs).tolist() == [1] * 40 + [2] * 10 + [3] * 30
    got = list(sh.run_stream(segs, batch_size=2))                       # batches in length order: [1, 2], then [0]
    assert [idx for idx, _ in got] == [[1, 2], [0]]
    assert [f.tolist() for f in got[0][1]] == [[2] * 10, [3] * 30]


def _gloo_worker(rank, world, port, q, mode):
    import os
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        segs = None
        lens = (4, 1, 3, 6, 2, 5, 7)
        if rank == 0:
            segs = [{"phones": [i + 1] * n, "bert_features": torch.zeros(1024, n), "norm_text": "x" * n} for i, n in enumerate(lens)]
            if mode == "bert":
                segs[2]["bert_features"] = torch.ones(1024, 3)
        calls = []

        def synth(segments):
            calls.append(len(segments))
            if mode == "slow_rank1" and rank == 1:
                time.sleep(0.4)                                     # uneven work: the queue must hand the rest to the idle ranks
            if mode == "fail" and rank == world - 1:
                raise RuntimeError("engine failure on the last rank")
            if mode in ("fail0_stream", "fail0_run") and rank == 0:
                time.sleep(0.2)                                     # the other rank has batches in flight when rank 0 fails
                raise RuntimeError("engine failure on rank 0")
            if mode == "many" and rank == 1:
                time.sleep(0.05)
            return _stub_synth(segments)
        sh = sharding.ShardedSynthesizer(synth, torch.device("cpu"), dynamic=mode not in ("static", "fail"))   # static dealing: the failing rank is sure to own a batch
        if mode == "many" and rank == 0:
            segs = [{"phones": [i % 50 + 1] * 2, "bert_features": None, "norm_text": "x" * (i + 1)} for i in range(24)]
        if mode == "equal" and rank == 0:       # equal lengths: every batch is a run of consecutive segments (configs[1]'s step)
            segs = [{"phones": [i + 1] * 3, "bert_features": None, "norm_text": "x" * 3} for i in range(8)]
        if mode in ("stream", "slow_rank1", "fail0_stream", "many"):
            order, flat = [], []
            try:
                for idxs, frags in sh.run_stream(segs, batch_size=2):
                    order.append(idxs)
                    flat.append([f.tolist() for f in frags])
                q.put((rank, {"order": order, "flat": flat, "calls": calls, "owner": list(sh.last_owner)}))
            except RuntimeError as e:
                q.put((rank, {"error": str(e), "order": order, "calls": calls}))
        else:
            try:
                out = sh.run(segs, batch_size=2 if mode != "weak" else None)
                q.put((rank, {"out": None if out is None else out.tolist(), "calls": calls}))
            except RuntimeError as e:
                q.put((rank, {"error": str(e), "calls": calls}))
    finally:
        dist.destroy_process_group()


def _run_gloo(world, mode, salt):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world * 7 + salt + (os_getpid() % 200)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


EXP = [1] * 40 + [2] * 10 + [3] * 30 + [4] * 60 + [5] * 20 + [6] * 50 + [7] * 70


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["dynamic", "static", "weak"])
def test_sharded_synthesizer_gloo_world(world, mode):
    """N > 1 on CPU: broadcast, batches from the work queue (store counter) or round-robin, point-to-point return of each
    batch, submission order restored; `weak` = one batch per rank (the benchmark's configs[1] step)."""
    res = _run_gloo(world, mode, {"dynamic": 0, "static": 11, "weak": 23}[mode])
    assert res[0]["out"] == EXP
    assert all(res[r]["out"] is None for r in range(1, world))
    assert sum(sum(res[r]["calls"]) for r in range(world)) == 7        # every utterance synthesised exactly once


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_streaming_yields_batches_in_order_with_work_queue(world):
    """BASELINE configs[4] semantics: rank 0 yields batch after batch in order (return_fragment, TTS.py:1321) while a slow
    rank holds one batch; the other ranks drain the queue meanwhile."""
    res = _run_gloo(world, "slow_rank1", 37)
    r0 = res[0]
    assert r0["order"] == [[1, 4], [2, 0], [5, 3], [6]]                # make_batches order of lens (4,1,3,6,2,5,7), size 2
    flat = sum(r0["flat"], [])
    assert flat == [[2] * 10, [5] * 20, [3] * 30, [1] * 40, [6] * 50, [4] * 60, [7] * 70]
    assert all(0 <= o < world for o in r0["owner"])                     # who served which batch is the queue's business
    assert sum(len(res[r]["calls"]) for r in range(world)) == 4
    assert all(res[r]["order"] == [] for r in range(1, world))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_places_whole_batches_of_consecutive_segments(world):
    """equal-length utterances (the benchmark step): each rank's batch is one run of consecutive segments and lands in the
    result buffer as ONE copy; the order must still be the submission order whoever served which batch."""
    res = _run_gloo(world, "equal", 97)
    assert res[0]["out"] == sum(([i + 1] * 30 for i in range(8)), [])
    assert sum(sum(res[r]["calls"]) for r in range(world)) == 8


def test_sharded_bert_features_are_shipped_and_failures_propagate():
    res = _run_gloo(2, "bert", 51)
    out = res[0]["out"]
    assert out[50:80] == [103] * 30 and out[:40] == [1] * 40           # the zh segment arrived WITH its features
    res = _run_gloo(3, "fail", 63)
    assert "error" in res[0] and "failed" in res[0]["error"]           # rank 0 raises instead of waiting for ever


@pytest.mark.parametrize("mode", ["fail0_stream", "fail0_run"])
def test_rank0_failure_drains_the_other_ranks_before_raising(mode):
    """rank 0's OWN batch fails while rank 1 has results in flight: rank 0 must receive them before it re-raises, otherwise
    rank 1 blocks in its send; every process exits with code 0 (checked by _run_gloo)."""
    res = _run_gloo(2, mode, {"fail0_stream": 71, "fail0_run": 83}[mode])
    assert "error" in res[0] and "rank 0" in res[0]["error"]
    assert "error" not in res[1]                                        # rank 1 finished its sends and left normally
    assert sum(res[1]["calls"]) >= 1


def test_streaming_many_batches_per_rank():
    """12 batches over 2 ranks (more than 2 per rank): all emitted in make_batches order, every batch once."""
    res = _run_gloo(2, "many", 97)
    r0 = res[0]
    assert r0["order"] == [[2 * i, 2 * i + 1] for i in range(12)]
    assert len(res[0]["calls"]) + len(res[1]["calls"]) == 12 and len(res[1]["calls"]) >= 1
    assert sorted(set(r0["owner"])) == [0, 1]


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


def test_tts_config_dict_yaml_and_defaults(tmp_path):
    """reference TTS_Config surface (TTS.py:217-410): dict with sections or custom keys, YAML path (created when missing),
    save / reload round trip, languages per version, equality by path."""
    from gsv.TTS_infer_pack.TTS import TTS_Config
    c = TTS_Config({"device": "cuda:0", "version": "v2", "max_batch": 4})
    assert c.version == "v2" and c.max_batch == 4 and c.is_half and not c.use_vocoder and "yue" in c.languages
    assert c.t2s_weights_path.endswith("s1bert25hz-5kh-longer-epoch=12-step=369668.ckpt")
    c2 = TTS_Config({"custom": {"device": "cuda:0", "is_half": False, "version": "v3", "t2s_weights_path": "a", "vits_weights_path": "b"}})
    assert c2.version == "v3" and not c2.is_half and c2.use_vocoder and c2.precision == torch.float32
    assert TTS_Config({"version": "v1"}).languages == ["auto", "en", "zh", "ja", "all_zh", "all_ja"]
    p = str(tmp_path / "cfg" / "tts_infer.yaml")
    c3 = TTS_Config(p)                                        # missing file: written from the defaults
    assert os.path.exists(p) and c3.version == "v2"
    c2.save_configs(p)
    c4 = TTS_Config(p)
    assert (c4.version, c4.is_half, c4.t2s_weights_path) == ("v3", False, "a") and c4 == TTS_Config(p) and hash(c4) == hash(p)
    assert set(c4.update_configs()) == {"device", "is_half", "version", "t2s_weights_path", "vits_weights_path", "bert_base_path",
                                        "cnhuhbert_base_path"}
    with pytest.raises(NotImplementedError):
        TTS_Config({"version": "v9"})
