"""TEST INFRASTRUCTURE ONLY -- inputs of the pipeline-glue fixtures (tests/golden/tts_glue_*.npz), shared by the generator
(oracle/gen_golden_tts_glue.py: runs the REFERENCE's TTS methods) and by the tests that compare the product / the oracle
with those fixtures.  Everything is regenerated from seeds (gsv.synthetic); nothing here is reference code."""
import torch

from gsv import synthetic as S

VC = {"T_ref": 20, "T_chunk": 48, "overlapped_len": 4}     # reduced so that short inputs span several chunks


def models(version):
    """(vits config, vits v3/v4 state dict incl. cfm.estimator.*, DiT config, vocoder config, vocoder state dict, vocoder kind)"""
    vcfg = S.small_vits_config()
    vcfg["model"]["inter_channels"] = vcfg["model"]["hidden_channels"]
    vcfg["model"]["version"] = version
    dit = S.small_dit_config()
    dit["text_dim"] = 512
    vsd = S.make_vits_v3_state_dict(vcfg, seed=12, dit_cfg=dit)
    kind = "bigvgan" if version == "v3" else "hifigan"
    ocfg = S.small_vocoder_config(kind)
    osd = S.make_vocoder_state_dict(ocfg, seed=13)
    return vcfg, vsd, dit, ocfg, osd, kind


def upsample_rate(ocfg):
    r = 1
    for u in ocfg["upsample_rates"]:
        r *= u
    return r


def prompt(Tm=26):
    refer = S.make_refer_spec(frames=30, seed=5)
    prompt_sem = torch.from_numpy(S.hash_ints("v3_prompt_sem", 8, 1024, 3))
    prompt_ph = S.hash_ints("v3_prompt_ph", 6, 732, 3).tolist()
    ref_mel = S.hash_symmetric("v3_ref_mel", (1, 100, Tm), 5.0, 3) - 5.0       # UN-normalised mel (what mel_fn returns)
    return refer, prompt_sem, prompt_ph, ref_mel


def noise_fn(call, shape):
    return S.hash_normal(f"v3_cfm_noise{call}", tuple(shape), 1)


def single_inputs():
    sem = torch.from_numpy(S.hash_ints("v3_sem", 19, 1024, 4)).view(1, 1, -1)
    ph = torch.from_numpy(S.hash_ints("v3_ph", 11, 732, 4)).view(1, -1)
    return sem, ph


def batched_inputs(case="ragged"):
    """`ragged`: the last chunk needs padding (the reference's slice `audio[ov*up : -pad*up]` is then non-empty);
    `exact`: the chunks divide evenly -> pad_len == 0 and the reference's slice is EMPTY (TTS.py:1600), reproduced as is."""
    if case == "ragged":
        lens_s, lens_p, idx = [9, 14, 6], [7, 9, 5], [9, 10, 6]
    else:
        lens_s, lens_p, idx = [12, 11], [7, 9], [12, 11]
    sems = [torch.from_numpy(S.hash_ints(f"v3_bsem{i}", n, 1024, 6)) for i, n in enumerate(lens_s)]
    phs = [torch.from_numpy(S.hash_ints(f"v3_bph{i}", n, 732, 6)) for i, n in enumerate(lens_p)]
    return idx, sems, phs


SOLA_CASES = [(2, 9000, 3072), (4, 700, 96), (3, 64, 8)]


def sola_fragments(n, length, ov):
    base = torch.cumsum(S.hash_symmetric("sola_base", (n * length + ov,), 1.0, 9), 0)
    base = base - torch.nn.functional.avg_pool1d(base.view(1, 1, -1), 201, 1, 100, count_include_pad=False).view(-1)
    base = base / base.abs().max()
    shifts = [0, 5, -7, 11]
    frags = []
    for i in range(n):
        s0 = i * (length - ov) + shifts[i % 4] * (i > 0)
        frags.append(base[max(s0, 0):max(s0, 0) + length].clone() * (1.0 + 0.05 * i))
    return frags


def postprocess_inputs(dtype=torch.float32):
    """three batches of ragged fragments; some exceed 1 in magnitude (peak-normalised), one is silent"""
    batch_index_list = [[4, 0, 6], [2, 5], [1, 3]]
    audio = []
    k = 0
    for bi, idxs in enumerate(batch_index_list):
        row = []
        for j, _ in enumerate(idxs):
            n = 300 + 137 * k
            amp = [0.4, 1.7, 0.0, 0.99, 2.5, 1.0, 0.05][k % 7]
            row.append((S.hash_symmetric(f"pp_frag{k}", (n,), 1.0, 2) * amp).to(dtype))
            k += 1
        audio.append(row)
    return audio, batch_index_list


TO_BATCH_CASES = [
    # (norm_text lengths, batch_size, threshold, split_bucket)
    ([12, 3, 40, 41, 39, 7, 8, 100, 5, 5, 5, 60], 4, 0.75, True),
    ([12, 3, 40, 41, 39, 7, 8, 100, 5, 5, 5, 60], 4, 0.75, False),
    ([1, 1, 1, 50, 1, 1, 1, 1], 8, 0.75, True),
    ([30] * 33, 32, 0.75, True),
    ([9], 5, 0.75, True),
    ([4, 90, 4, 90, 4, 90, 5, 80], 3, 0.95, True),
    (list(range(1, 41)), 6, 0.5, True),
]


def to_batch_data(lens):
    data = []
    for i, n in enumerate(lens):
        nph = 2 + (n * 7 + i) % 11
        data.append({"phones": S.hash_ints(f"tb_ph{i}", nph, 732, 8).tolist(), "bert_features": torch.zeros(1024, nph),
                     "norm_text": "x" * n})
    prompt_data = {"phones": S.hash_ints("tb_prompt", 6, 732, 8).tolist(), "bert_features": torch.zeros(1024, 6), "norm_text": "pppp"}
    return data, prompt_data
