"""TEST INFRASTRUCTURE ONLY -- the parity cases (model size, seeds, sampling settings) shared by
oracle/gen_golden*.py (which runs the reference on them) and tests/ (which run the oracle and the
HIP path on them).  Inputs are regenerated from seeds, so fixtures only hold expected outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
from gsv import synthetic as S  # noqa: E402

# ---------------------------------------------------------------------------------
# T2S cases: (name, config kwargs, batch spec, sampling)
# ---------------------------------------------------------------------------------
T2S_CASES = {
    # reduced model, ragged batch, greedy, EOS reachable (vocab 65)
    "t2s_small_greedy": dict(cfg=("small", dict(n_layer=2, dim=128, head=4, vocab=65, phoneme_vocab=48)), seed=3,
                             x_lens=[9, 14, 6], P=7, top_k=1, top_p=1.0, temperature=1.0, rep=1.35,
                             early_stop=40, shared_noise=False),
    # reduced model, stochastic sampling with injected Exp(1) noise, top-k + top-p + temperature
    "t2s_small_sample": dict(cfg=("small", dict(n_layer=2, dim=128, head=4, vocab=65, phoneme_vocab=48)), seed=5,
                             x_lens=[11, 5, 8, 13], P=6, top_k=5, top_p=0.9, temperature=0.8, rep=1.35,
                             early_stop=30, shared_noise=True),
    # reduced model, top_k only (TTS.run defaults top_k=5, top_p=1)
    "t2s_small_topk": dict(cfg=("small", dict(n_layer=3, dim=64, head=2, vocab=33, phoneme_vocab=48)), seed=7,
                           x_lens=[10, 10], P=5, top_k=5, top_p=1.0, temperature=1.0, rep=1.35,
                           early_stop=25, shared_noise=True),
    # full v2 architecture, ragged B=2, greedy
    "t2s_v2_greedy": dict(cfg=("v2", {}), seed=0, x_lens=[20, 13], P=12, top_k=1, top_p=1.0, temperature=1.0,
                          rep=1.35, early_stop=24, shared_noise=False),
}


def t2s_case_inputs(case):
    kind, kw = case["cfg"]
    cfg = S.small_t2s_config(**kw) if kind == "small" else S.T2S_V2_CONFIG
    m = cfg["model"]
    sd = S.make_t2s_state_dict(cfg, seed=case["seed"])
    xs = [torch.from_numpy(S.hash_ints(f"x{i}", n, m["phoneme_vocab_size"], case["seed"])).long()
          for i, n in enumerate(case["x_lens"])]
    berts = [S.hash_symmetric(f"bert{i}", (1024, n), 0.5, case["seed"]) for i, n in enumerate(case["x_lens"])]
    prompt = torch.from_numpy(S.hash_ints("prompt", case["P"], m["vocab_size"] - 1, case["seed"])).long()
    prompts = prompt.unsqueeze(0).expand(len(xs), -1).contiguous()
    noise = None
    if case["shared_noise"]:
        u = S.hash_uniform("expnoise", 1500 * m["vocab_size"], case["seed"]).astype(np.float64)
        noise = torch.from_numpy((-np.log1p(-u)).astype(np.float32).reshape(1500, 1, m["vocab_size"]))
        noise = noise.clamp_min(1e-10)
    return cfg, sd, xs, berts, prompts, noise



VITS_CASES = {
    "vits_small": dict(cfg="small", seed=2, T=12, L=9, Tr=[30], noise_scale=0.5),
    "vits_small_2ref": dict(cfg="small", seed=4, T=7, L=5, Tr=[24, 17], noise_scale=0.5),
    "vits_v2": dict(cfg="v2", seed=0, T=10, L=8, Tr=[40], noise_scale=0.5),
    # speed != 1: linear interpolation of the encoder output (reference models.py:226-228)
    "vits_small_speed": dict(cfg="small", seed=6, T=11, L=7, Tr=[21], noise_scale=0.5, speed=1.3),
    # v2Pro conditioning (N4): gin 1024, ge += sv_emb(sv); PReLU; MRTE gets ge_to512(ge) (models.py:895-899, 971-975, 997)
    "vits_small_v2pro": dict(cfg="small", seed=8, T=9, L=6, Tr=[22, 15], noise_scale=0.5, version="v2Pro"),
}


def vits_case_inputs(case):
    cfg = S.small_vits_config() if case["cfg"] == "small" else S.VITS_V2_CONFIG
    if case.get("version") in ("v2Pro", "v2ProPlus"):
        import copy
        cfg = copy.deepcopy(cfg)
        cfg["model"]["version"] = case["version"]
        cfg["model"]["gin_channels"] = 1024              # reference configs/s2v2Pro.json
    sd = S.make_vits_state_dict(cfg, seed=case["seed"])
    codes = torch.from_numpy(S.hash_ints("codes", case["T"], 1024, case["seed"])).view(1, 1, -1)
    text = torch.from_numpy(S.hash_ints("text", case["L"], cfg["n_symbols"], case["seed"])).view(1, -1)
    refers = [torch.from_numpy(S.hash_uniform(f"refer{i}", 1025 * tr, case["seed"]).reshape(1, 1025, tr).copy())
              for i, tr in enumerate(case["Tr"])]
    speed = case.get("speed", 1)
    frames = 2 * case["T"] if speed == 1 else int(2 * case["T"] / speed) + 1
    noise = S.hash_normal("vits_noise", (cfg["model"]["inter_channels"], frames), case["seed"])
    ssl = S.hash_symmetric("ssl", (1, 768, 2 * case["T"]), 1.0, case["seed"])
    return cfg, sd, codes, text, refers, noise, ssl


def vits_case_sv_emb(case):
    """speaker-verification embeddings (one [1, 20480] per reference) of a v2Pro case, else None"""
    if case.get("version") not in ("v2Pro", "v2ProPlus"):
        return None
    return [S.hash_symmetric(f"sv_emb{i}", (1, 20480), 1.0, case["seed"]) for i in range(len(case["Tr"]))]




VOC_CASES = {
    "voc_hifigan_small": dict(kind="hifigan", small=True, seed=1, F=23),
    "voc_bigvgan_small": dict(kind="bigvgan", small=True, seed=2, F=19),
    "voc_hifigan_v4": dict(kind="hifigan", small=False, seed=3, F=12),
    "voc_bigvgan_v2": dict(kind="bigvgan", small=False, seed=4, F=6),
}


def voc_case_inputs(case):
    if case["small"]:
        cfg = S.small_vocoder_config(case["kind"])
    else:
        cfg = dict(S.HIFIGAN_V4_CONFIG if case["kind"] == "hifigan" else S.BIGVGAN_V2_24K_CONFIG)
    sd = S.make_vocoder_state_dict(cfg, seed=case["seed"])
    mel = S.hash_symmetric("voc_mel", (1, 100, case["F"]), 2.0, case["seed"])
    return cfg, sd, mel


CFM_CASES = {
    "cfm_small": dict(cfg="small", seed=1, B=1, T=40, Tp=12, steps=4),
    "cfm_small_b2": dict(cfg="small", seed=2, B=2, T=33, Tp=9, steps=3),
    "cfm_v3dims": dict(cfg="v3x2", seed=3, B=1, T=36, Tp=10, steps=2),      # v3 widths, depth 2
}


def cfm_case_inputs(case):
    if case["cfg"] == "small":
        cfg = S.small_dit_config()
    else:
        cfg = dict(S.DIT_V3_CONFIG)
        cfg["depth"] = 2
    sd = S.make_dit_state_dict(cfg, seed=case["seed"])
    B, T, Tp = case["B"], case["T"], case["Tp"]
    mu = S.hash_symmetric("cfm_mu", (B, T, cfg["text_dim"]), 1.0, case["seed"])
    prompt = S.hash_symmetric("cfm_prompt", (B, cfg["mel_dim"], Tp), 1.0, case["seed"])
    noise = S.hash_normal("cfm_noise", (B, cfg["mel_dim"], T), case["seed"])
    return cfg, sd, mu, prompt, noise


ENCP_CASES = {
    "encp_small_v3": dict(version="v3", seed=1, T=21, L=13, Tr=30, speed=1),
    "encp_small_v4": dict(version="v4", seed=2, T=17, L=9, Tr=24, speed=1),
    "encp_small_v3_speed": dict(version="v3", seed=3, T=3, L=5, Tr=20, speed=1.2),      # 1 masked tail frame
    "encp_small_v4_speed": dict(version="v4", seed=4, T=9, L=7, Tr=20, speed=0.8),
}


def encp_case_inputs(case):
    cfg = S.small_vits_config()
    cfg["model"]["inter_channels"] = cfg["model"]["hidden_channels"]     # bridge is Conv1d(inter, 512) applied to the hidden sequence
    sd = S.make_vits_v3_state_dict(cfg, seed=case["seed"])
    codes = torch.from_numpy(S.hash_ints("codes", case["T"], 1024, case["seed"])).view(1, 1, -1)
    text = torch.from_numpy(S.hash_ints("text", case["L"], cfg["n_symbols"], case["seed"])).view(1, -1)
    refer = torch.from_numpy(S.hash_uniform("refer0", 1025 * case["Tr"], case["seed"]).reshape(1, 1025, case["Tr"]).copy())
    return cfg, sd, codes, text, refer
