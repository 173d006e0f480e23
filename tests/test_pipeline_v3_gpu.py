"""GPU: the v3 / v4 synthesis path (H14 + H15/H16 + H17) through the TTS mirror -- decode_encp -> chunked CFM/DiT ->
vocoder (-> SOLA for the batched variant) -- against the same chain built from the oracles (oracle/tts_v3_oracle.py)
on reduced models in fp32, with the CFM noise pinned; and `gsv_sola` alone against the oracle's sola_algorithm."""
import numpy as np
import pytest
import torch

from oracle import cfm_oracle, tts_v3_oracle, vocoder_oracle
from oracle.vits_oracle import VitsOracle
from gsv import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VC = {"T_ref": 20, "T_chunk": 48, "overlapped_len": 4}     # reduced so that short test inputs span several chunks


def _build(version, is_half=False):
    from gsv.TTS_infer_pack.TTS import TTS
    tcfg = S.small_t2s_config(n_layer=2, dim=128, head=4, vocab=1025, phoneme_vocab=732)
    tcfg["data"]["max_sec"] = 0.4
    tsd = S.make_t2s_state_dict(tcfg, seed=11, suppress_eos=True)
    vcfg = S.small_vits_config()
    vcfg["model"]["inter_channels"] = vcfg["model"]["hidden_channels"]
    vcfg["model"]["version"] = version
    dit = S.small_dit_config()
    dit["text_dim"] = 512
    vsd = S.make_vits_v3_state_dict(vcfg, seed=12, dit_cfg=dit)
    vcfg["dit"] = {k: v for k, v in dit.items() if k != "mel_dim"}
    kind = "bigvgan" if version == "v3" else "hifigan"
    ocfg = S.small_vocoder_config(kind)
    osd = S.make_vocoder_state_dict(ocfg, seed=13)
    tts = TTS({"device": DEV, "is_half": is_half, "version": version, "max_batch": 4, "max_seq": 256})
    tts.init_t2s_weights(state={"weight": tsd, "config": tcfg})
    tts.init_vits_weights(state={"weight": vsd, "config": vcfg})
    tts.init_vocoder(state={"weight": osd, "config": dict(ocfg, **VC)})
    return tts, (tcfg, tsd), (vcfg, vsd, dit), (ocfg, osd, kind)


def _prompt(tts, Tm=26):
    refer = S.make_refer_spec(frames=30, seed=5)
    prompt_sem = torch.from_numpy(S.hash_ints("v3_prompt_sem", 8, 1024, 3))
    prompt_ph = S.hash_ints("v3_prompt_ph", 6, 732, 3).tolist()
    ref_mel = S.hash_symmetric("v3_ref_mel", (1, 100, Tm), 5.0, 3) - 5.0
    tts.set_prompt_cache(prompt_sem, [refer.to(DEV)], phones=prompt_ph, bert_features=torch.zeros(1024, 6), norm_text="xxxxxx",
                         ref_mel=ref_mel)
    return refer, prompt_sem, torch.tensor(prompt_ph), ref_mel


def _oracle_stages(vits, osd_cfg, version, noise_fn):
    vcfg, vsd, dit = vits
    ocfg, osd, kind = osd_cfg
    vo = VitsOracle(vsd, vcfg)
    dsd = {k[len("cfm.estimator."):]: v for k, v in vsd.items() if k.startswith("cfm.estimator.")}

    def decode_encp(codes, text, refer, ge, speed):
        return vo.decode_encp(codes, text, refer, speed=speed, version=version)

    def cfm(fea, mel2, steps, call):
        nz = noise_fn(call, (fea.shape[0], 100, fea.shape[1]))
        prompt = mel2.expand(fea.shape[0], -1, -1)
        return cfm_oracle.cfm_inference(dsd, dit, fea, prompt, steps, nz.clone())

    def voc(mel):
        return (vocoder_oracle.bigvgan if kind == "bigvgan" else vocoder_oracle.hifigan)(osd, ocfg, mel)

    return decode_encp, cfm, voc


def _noise_fn(call, shape):
    return S.hash_normal(f"v3_cfm_noise{call}", shape, 1)


@pytest.mark.parametrize("version", ["v3", "v4"])
def test_using_vocoder_synthesis_matches_oracle_chain(version):
    """one fragment, mel generated in 3 chunks each prompted by the previous tail; fp32: waveform max-abs <= 5e-3."""
    tts, t2s, vits, voc = _build(version)
    refer, psem, pph, ref_mel = _prompt(tts)
    sem = torch.from_numpy(S.hash_ints("v3_sem", 19, 1024, 4)).view(1, 1, -1)
    ph = torch.from_numpy(S.hash_ints("v3_ph", 11, 732, 4)).view(1, -1)
    wav = tts.using_vocoder_synthesis(sem.to(DEV), ph.to(DEV), speed=1.0, sample_steps=3, noise_fn=_noise_fn).float().cpu()
    de, cfm, vo = _oracle_stages(vits, voc, version, _noise_fn)
    vc = dict(tts.vocoder_configs)
    ref = tts_v3_oracle.using_vocoder_synthesis(de, cfm, vo, vc, psem, pph, refer, ref_mel, sem, ph, 1.0, 3)
    assert wav.shape == ref.shape
    assert wav.shape[0] == (int(2 * 19 * 1.875) if version == "v3" else 2 * 19 * 2) * vc["upsample_rate"]
    assert (wav - ref).abs().max() <= 5e-3


def test_batched_infer_with_sola_matches_oracle_chain():
    """three fragments -> overlapping chunks -> ONE batched CFM call -> vocoder -> SOLA -> split (TTS.py:1496-1609)."""
    tts, t2s, vits, voc = _build("v3")
    refer, psem, pph, ref_mel = _prompt(tts)
    sems = [torch.from_numpy(S.hash_ints(f"v3_bsem{i}", n, 1024, 6)) for i, n in enumerate([9, 14, 6])]
    phs = [torch.from_numpy(S.hash_ints(f"v3_bph{i}", n, 732, 6)) for i, n in enumerate([7, 9, 5])]
    idx = [9, 10, 6]                                           # the second fragment keeps only its last 10 tokens
    out = tts.using_vocoder_synthesis_batched_infer(idx, [s.to(DEV) for s in sems], [p.to(DEV) for p in phs], speed=1.0,
                                                    sample_steps=2, noise_fn=_noise_fn)
    de, cfm, vo = _oracle_stages(vits, voc, "v3", _noise_fn)
    ref = tts_v3_oracle.using_vocoder_synthesis_batched_infer(de, cfm, vo, dict(tts.vocoder_configs), psem, pph, refer, ref_mel,
                                                              idx, sems, phs, 1.0, 2)
    assert len(out) == len(ref) == 3
    for a, b in zip(out, ref):
        assert a.shape == b.shape
        if b.numel():
            assert (a.float().cpu() - b).abs().max() <= 5e-3
    assert sum(int(b.numel()) for b in ref) > 0


@pytest.mark.parametrize("n,length,ov", [(2, 9000, 3072), (4, 700, 96), (1, 50, 8)])
def test_sola_kernel_matches_oracle(n, length, ov):
    """gsv_sola vs the restated sola_algorithm on smooth fragments with a clear correlation peak: same stitched length
    (= same argmax offsets) and samples within 1e-5."""
    from gsv.TTS_infer_pack.TTS import TTS
    tts = TTS({"device": DEV, "is_half": False, "version": "v3"})
    base = torch.cumsum(S.hash_symmetric("sola_base", (n * length + ov,), 1.0, 9), 0)
    base = base - torch.nn.functional.avg_pool1d(base.view(1, 1, -1), 201, 1, 100, count_include_pad=False).view(-1)
    base = base / base.abs().max()
    shifts = [0, 5, -7, 11]
    frags = []
    for i in range(n):
        s0 = i * (length - ov) + shifts[i % 4] * (i > 0)
        frags.append(base[max(s0, 0):max(s0, 0) + length].clone() * (1.0 + 0.05 * i))
    ref = tts_v3_oracle.sola_algorithm(frags, ov) if n > 1 else frags[0]
    out = tts.sola_algorithm([f.to(DEV) for f in frags], ov).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max() <= 1e-5


def test_tts_run_v3_end_to_end():
    """AR -> v3 path through TTS.run (parallel and per-fragment variants): sample rate, dtype, non-silent output."""
    tts, *_ = _build("v3")
    _prompt(tts)
    utt = S.make_utterances(3, prompt_phones=6, target_phones=9, prompt_tokens=8, seed=21, ragged=True)
    segs = [{"phones": it["phones"], "bert_features": torch.zeros(1024, len(it["phones"])), "norm_text": "x" * (4 + i)}
            for i, it in enumerate(utt["items"])]
    for par in (True, False):
        out = list(tts.run({"segments": segs, "batch_size": 3, "top_k": 1, "seed": 3, "parallel_infer": par, "sample_steps": 2,
                            "fragment_interval": 0.01}))
        assert len(out) == 1
        sr, audio = out[0]
        assert sr == 24000 and audio.dtype == np.int16 and audio.size > 0
        if not par:
            assert np.abs(audio).max() > 0


@pytest.mark.parametrize("version", ["v3", "v4"])
def test_ref_mel_from_raw_audio_feeds_the_prompt(version):
    """TTS.py:1442-1453: without an explicit ref_mel the prompt mel comes from the cached reference audio -- stereo mixed to
    mono, resampled to 24 kHz (v3) / 32 kHz (v4), mel_fn / mel_fn_v4, norm_spec, cut to T_min and the last T_ref frames --
    against the oracle mel of the same resampled samples (the resampler itself is parity-unpinned, see gsv/audio_io.py)."""
    from gsv.audio_io import resample
    from gsv.TTS_infer_pack.TTS import norm_spec
    from oracle import mel_filterbank as omf
    tts, *_ = _build(version)
    refer, prompt_sem, prompt_ph, _ = _prompt(tts)
    tts.prompt_cache["ref_mel"] = None
    with pytest.raises(Exception):
        tts._prompt_features()                                   # neither ref_mel nor reference audio
    sr0 = 32000 if version == "v3" else 24000                    # both versions resample
    stereo = torch.stack([S.make_waveform(sr0 // 2, 1, sr=sr0), S.make_waveform(sr0 // 2, 2, sr=sr0)]).numpy()
    tts.prompt_cache["raw_audio"], tts.prompt_cache["raw_sr"] = stereo, sr0
    spec, fea_ref, ge, mel2, T_min = tts._prompt_features()
    tgt = 24000 if version == "v3" else 32000
    mono = torch.from_numpy(resample(stereo.mean(0, keepdims=True), sr0, tgt))
    n_fft, hop = (1024, 256) if version == "v3" else (1280, 320)
    ref = norm_spec(omf.mel_spectrogram(mono, n_fft, 100, tgt, hop, n_fft, 0, None))
    T_ref = VC["T_ref"]
    full = int(2 * 8 * 1.875) if version == "v3" else 2 * 8 * 2  # fea_ref frames of the 8 prompt tokens before any cut
    assert ref.shape[2] > full > T_ref and T_min == T_ref and mel2.shape == (1, 100, T_min) and fea_ref.shape[2] == T_min
    want = ref[:, :, :full]
    want = want[:, :, -T_ref:] if want.shape[2] > T_ref else want
    assert (mel2.float().cpu() - want).abs().max() <= 5e-4


def test_lora_checkpoint_is_merged_into_the_base_model():
    """init_vits_weights with a v3 LoRA state ("lora_rank" + peft-named lora_A / lora_B, TTS.py:556-572): the engine is built from
    W + B A -- same CFM output as an engine given the merged weights directly, different from the base model's; a missing base
    model raises FileExistsError (TTS.py:491-493)."""
    from gsv import process_ckpt as pc
    tts, _, (vcfg, vsd, dit), _ = _build("v3")
    lw = S.make_lora_state_dict(vsd, rank=4, seed=3)
    merged = pc.merge_lora_v3(vsd, lw, 4)

    def cfm_out(t):
        fea = S.hash_symmetric("lora_fea", (1, 40, 512), 1.0, 1).to(DEV)
        mel = S.hash_symmetric("lora_mel", (1, 100, 16), 1.0, 2).to(DEV)
        return t.vits_model.cfm.inference(fea, torch.LongTensor([40]).to(DEV), mel, 2, inference_cfg_rate=0,
                                          noise=S.hash_normal("lora_noise", (1, 100, 40), 1).to(DEV)).float().cpu()

    base_out = cfm_out(tts)
    tts.init_vits_weights(state={"weight": lw, "config": vcfg, "lora_rank": 4}, base_state={"weight": vsd, "config": vcfg})
    lora_out = cfm_out(tts)
    tts.init_vits_weights(state={"weight": merged, "config": vcfg})
    want = cfm_out(tts)
    assert torch.equal(lora_out, want) and (lora_out - base_out).abs().max() > 1e-3
    with pytest.raises(FileExistsError):
        tts.init_vits_weights(state={"weight": lw, "config": vcfg, "lora_rank": 4})


# ---------------------------------------------------------------------------------------------------------------------
# against fixtures written by the REFERENCE's own TTS methods (oracle/gen_golden_tts_glue.py): H17 is pinned
# ---------------------------------------------------------------------------------------------------------------------
def _build_glue(version):
    from gsv.TTS_infer_pack.TTS import TTS
    from oracle import glue_cases as G
    tcfg = S.small_t2s_config(n_layer=2, dim=128, head=4, vocab=1025, phoneme_vocab=732)
    tsd = S.make_t2s_state_dict(tcfg, seed=11, suppress_eos=True)
    vcfg, vsd, dit, ocfg, osd, kind = G.models(version)
    vcfg = dict(vcfg, dit={k: v for k, v in dit.items() if k != "mel_dim"})
    tts = TTS({"device": DEV, "is_half": False, "version": version, "max_batch": 4, "max_seq": 256})
    tts.init_t2s_weights(state={"weight": tsd, "config": tcfg})
    tts.init_vits_weights(state={"weight": vsd, "config": vcfg})
    tts.init_vocoder(state={"weight": osd, "config": dict(ocfg, **G.VC)})
    refer, psem, pph, ref_mel = G.prompt()
    tts.set_prompt_cache(psem, [refer.to(DEV)], phones=pph, bert_features=torch.zeros(1024, 6), norm_text="xxxxxx", ref_mel=ref_mel)
    return tts


@pytest.mark.parametrize("version", ["v3", "v4"])
def test_v3_glue_matches_fixtures_of_the_reference_methods(version):
    """TTS.using_vocoder_synthesis and ..._batched_infer (HIP engines, fp32) == what the reference's own methods returned
    over the reference's own stage classes: single fragment in 3 prompted chunks, and the batched variant (overlapping
    chunks -> one batched CFM -> vocoder -> SOLA -> split) with equal fragment lengths (= equal SOLA offsets)."""
    from conftest import load_golden
    from oracle import glue_cases as G
    g = load_golden(f"tts_glue_{version}")
    tts = _build_glue(version)
    sem, ph = G.single_inputs()
    wav = tts.using_vocoder_synthesis(sem.to(DEV), ph.to(DEV), speed=1.0, sample_steps=3, noise_fn=G.noise_fn).float().cpu().numpy()
    assert wav.shape == g["single"].shape
    err = np.abs(wav - g["single"]).max()
    print(f"[glue] {version} single fragment vs the reference method: max-abs {err:.2e}")
    assert err <= 5e-3
    for case in ("ragged", "exact"):
        idx, sems, phs = G.batched_inputs(case)
        out = tts.using_vocoder_synthesis_batched_infer(idx, [s.to(DEV) for s in sems], [p.to(DEV) for p in phs], speed=1.0,
                                                        sample_steps=2, noise_fn=G.noise_fn)
        assert [int(o.numel()) for o in out] == g[f"batched_{case}_lens"].tolist()
        err = np.abs(torch.cat([o.float().cpu() for o in out]).numpy() - g[f"batched_{case}"]).max()
        print(f"[glue] {version} batched {case} vs the reference method: max-abs {err:.2e}")
        assert err <= 5e-3


def test_sola_and_postprocess_kernels_match_fixtures_of_the_reference_methods():
    """gsv_sola vs TTS.sola_algorithm and gsv_postprocess vs TTS.audio_postprocess (peak division in the fragment dtype,
    silence gap, recovery order, x 32768 -> int16), both as the reference's own methods computed them."""
    from conftest import load_golden
    from gsv.TTS_infer_pack.TTS import TTS
    from oracle import glue_cases as G
    g = load_golden("tts_glue_host")
    tts = TTS({"device": DEV, "is_half": False, "version": "v2"})
    for ci, (n, length, ov) in enumerate(G.SOLA_CASES):
        out = tts.sola_algorithm([f.to(DEV) for f in G.sola_fragments(n, length, ov)], ov).cpu().numpy()
        assert out.shape == g[f"sola{ci}"].shape and np.abs(out - g[f"sola{ci}"]).max() <= 1e-5
    for name, dtype in (("f32", torch.float32), ("f16", torch.float16)):
        t = TTS({"device": DEV, "is_half": dtype == torch.float16, "version": "v2"})
        for sb in (True, False):
            audio, bil = G.postprocess_inputs(dtype)
            sr, a16 = t.audio_postprocess([[f.to(DEV) for f in row] for row in audio], 32000, bil, 1.0, sb, 0.3)
            want = g[f"post_{name}_{'bucket' if sb else 'flat'}"]
            assert sr == 32000 and a16.dtype == np.int16 and a16.shape == want.shape
            assert np.array_equal(a16, want), f"{name} split_bucket={sb}: {int((a16 != want).sum())} samples differ"
