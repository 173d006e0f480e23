"""Deterministic, repo-owned synthetic checkpoints and inputs.

No pretrained weights exist offline, so parity tests, the smoke
test and bench.py all run on weights produced here: a counter hash (SplitMix64)
-> uniform -> fan-in scaled values, computed with numpy integer arithmetic only,
so the build container and the GPU box produce bit-identical tensors.  The
state-dict key names and shapes are the reference's checkpoint schema
(reference AR/models/t2s_model.py:260-351,
module/models.py:796-899), i.e. what `TTS.init_t2s_weights` / `init_vits_weights`
(reference TTS_infer_pack/TTS.py:484-594) would hand to `load_state_dict`.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

# v2 hyper-parameters (reference GPT_SoVITS/configs/s1longer-v2.yaml:19-28).
T2S_V2_CONFIG = {
    "model": {
        "vocab_size": 1025,
        "phoneme_vocab_size": 732,
        "embedding_dim": 512,
        "hidden_dim": 512,
        "head": 16,
        "linear_units": 2048,
        "n_layer": 24,
        "dropout": 0,
        "EOS": 1024,
        "random_bert": 0,
    },
    "data": {"max_sec": 54, "pad_val": 1024},
}

# v2 SoVITS hyper-parameters (reference GPT_SoVITS/configs/s2.json:22-88).
VITS_V2_CONFIG = {
    "data": {
        "sampling_rate": 32000,
        "filter_length": 2048,
        "hop_length": 640,
        "win_length": 2048,
        "n_speakers": 300,
    },
    "train": {"segment_size": 20480},
    "model": {
        "inter_channels": 192,
        "hidden_channels": 192,
        "filter_channels": 768,
        "n_heads": 2,
        "n_layers": 6,
        "kernel_size": 3,
        "p_dropout": 0.1,
        "resblock": "1",
        "resblock_kernel_sizes": [3, 7, 11],
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "upsample_rates": [10, 8, 2, 2, 2],
        "upsample_initial_channel": 512,
        "upsample_kernel_sizes": [16, 16, 8, 2, 2],
        "n_layers_q": 3,
        "use_spectral_norm": False,
        "gin_channels": 512,
        "semantic_frame_rate": "25hz",
        "freeze_quantizer": True,
    },
    "version": "v2",
    "n_symbols": 732,  # len(text/symbols2.py symbols), reference text/symbols2.py:783-788
}


def small_t2s_config(n_layer=2, dim=128, head=4, vocab=65, phoneme_vocab=48):
    """Reduced GPT config for unit tests (same code path, KB-sized fixtures)."""
    return {
        "model": {
            "vocab_size": vocab,
            "phoneme_vocab_size": phoneme_vocab,
            "embedding_dim": dim,
            "hidden_dim": dim,
            "head": head,
            "linear_units": dim * 4,
            "n_layer": n_layer,
            "dropout": 0,
            "EOS": vocab - 1,
            "random_bert": 0,
        },
        "data": {"max_sec": 54, "pad_val": vocab - 1},
    }


def small_vits_config():
    """Reduced SoVITS config for unit tests: same topology, fewer channels/layers.
    Only hps-controlled sizes shrink, so the reference class can still be built from it."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in VITS_V2_CONFIG.items()}
    cfg["model"].update(
        {
            "inter_channels": 64,
            "hidden_channels": 192,   # fixed by MRTE()'s defaults (reference mrte_model.py:11-15)
            "filter_channels": 256,
            "n_layers": 2,
            "upsample_rates": [4, 2, 2],
            "upsample_initial_channel": 128,
            "upsample_kernel_sizes": [8, 4, 2],
        }
    )
    return cfg


# --------------------------------------------------------------------------
# counter-hash RNG
# --------------------------------------------------------------------------
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [0, 1), a pure function of (name, seed, index)."""
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _splitmix64(ctr) >> np.uint64(40)  # 24 bits
    return (bits.astype(np.float32)) * np.float32(1.0 / (1 << 24))


def hash_symmetric(name: str, shape, amp: float, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n, seed)
    v = (u * np.float32(2.0) - np.float32(1.0)) * np.float32(amp)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def hash_normal(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Standard normal via Box-Muller on two hash streams (float32, deterministic)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = hash_uniform(name + "#a", n, seed).astype(np.float64)
    u2 = hash_uniform(name + "#b", n, seed).astype(np.float64)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32).reshape(shape))


def hash_ints(name: str, n: int, hi: int, seed: int = 0) -> np.ndarray:
    u = hash_uniform(name, n, seed)
    return np.minimum((u * hi).astype(np.int64), hi - 1)


def _w(name, shape, fan_in, gain=1.0, seed=0):
    return hash_symmetric(name, shape, gain * math.sqrt(3.0 / max(fan_in, 1)), seed)


def _b(name, n, amp=0.05, seed=0):
    return hash_symmetric(name, (n,), amp, seed)


# --------------------------------------------------------------------------
# GPT (Text2SemanticDecoder) checkpoint
# --------------------------------------------------------------------------
def make_t2s_state_dict(config=None, seed: int = 0, suppress_eos: bool = False,
                        logit_gain: float = 4.0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 state dict with the reference's key names *without* the Lightning
    `model.` prefix (reference t2s_lightning_module.py:23 adds it; TTS.py:590-594)."""
    cfg = (config or T2S_V2_CONFIG)["model"]
    d = cfg["hidden_dim"]
    ff = cfg.get("linear_units", 4 * d)
    # reference t2s_model.py:304 uses dim_feedforward = 4*d regardless of linear_units
    ff = 4 * d
    V = cfg["vocab_size"]
    PV = cfg["phoneme_vocab_size"]
    L = cfg["n_layer"]
    sd = OrderedDict()
    sd["bert_proj.weight"] = _w("bert_proj.weight", (d, 1024), 1024, 1.0, seed)
    sd["bert_proj.bias"] = _b("bert_proj.bias", d, 0.05, seed)
    sd["ar_text_embedding.word_embeddings.weight"] = hash_symmetric("ar_text_embedding", (PV, d), 1.0, seed)
    sd["ar_text_position.alpha"] = torch.tensor([0.8], dtype=torch.float32)
    sd["ar_audio_embedding.word_embeddings.weight"] = hash_symmetric("ar_audio_embedding", (V, d), 1.0, seed)
    sd["ar_audio_position.alpha"] = torch.tensor([1.2], dtype=torch.float32)
    for i in range(L):
        p = f"h.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = _w(p + "in_proj_weight", (3 * d, d), d, 1.0, seed)
        sd[p + "self_attn.in_proj_bias"] = _b(p + "in_proj_bias", 3 * d, 0.05, seed)
        sd[p + "self_attn.out_proj.weight"] = _w(p + "out_proj.weight", (d, d), d, 1.0, seed)
        sd[p + "self_attn.out_proj.bias"] = _b(p + "out_proj.bias", d, 0.05, seed)
        sd[p + "linear1.weight"] = _w(p + "linear1.weight", (ff, d), d, 1.0, seed)
        sd[p + "linear1.bias"] = _b(p + "linear1.bias", ff, 0.05, seed)
        sd[p + "linear2.weight"] = _w(p + "linear2.weight", (d, ff), ff, 1.0, seed)
        sd[p + "linear2.bias"] = _b(p + "linear2.bias", d, 0.05, seed)
        sd[p + "norm1.weight"] = 1.0 + hash_symmetric(p + "norm1.weight", (d,), 0.1, seed)
        sd[p + "norm1.bias"] = _b(p + "norm1.bias", d, 0.1, seed)
        sd[p + "norm2.weight"] = 1.0 + hash_symmetric(p + "norm2.weight", (d,), 0.1, seed)
        sd[p + "norm2.bias"] = _b(p + "norm2.bias", d, 0.1, seed)
    sd["ar_predict_layer.weight"] = _w("ar_predict_layer.weight", (V, d), d, logit_gain, seed)
    if suppress_eos:
        # Fixed-length synthetic workloads (bench.py): make EOS unreachable by giving the
        # last LayerNorm a constant offset and the EOS row a constant negative weight, so
        # logit[EOS] ~= -8 while the other logits have std ~ logit_gain.
        p = f"h.layers.{L - 1}."
        sd[p + "norm2.bias"] = torch.full((d,), 0.5, dtype=torch.float32)
        sd["ar_predict_layer.weight"][V - 1] = -16.0 / d
    return sd


# --------------------------------------------------------------------------
# SoVITS v2 (SynthesizerTrn) checkpoint, inference keys only (enc_q is dropped at
# export: reference process_ckpt.py:45-48)
# --------------------------------------------------------------------------
def _wn(sd, prefix, shape, fan_in, gain, seed, bias=True, bias_n=None):
    """weight-normed conv: store weight_g / weight_v like torch.nn.utils.weight_norm
    (dim=0), reference module/modules.py:159-176, module/models.py:427-437."""
    v = _w(prefix + ".weight_v", shape, fan_in, gain, seed)
    norm = v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], *([1] * (len(shape) - 1)))
    g = norm * (1.0 + hash_symmetric(prefix + ".weight_g", norm.shape, 0.1, seed))
    sd[prefix + ".weight_g"] = g
    sd[prefix + ".weight_v"] = v
    if bias:
        sd[prefix + ".bias"] = _b(prefix + ".bias", bias_n if bias_n is not None else shape[0], 0.05, seed)


def _encoder(sd, prefix, n_layers, hidden, filt, n_heads, ksize, seed, window=4):
    kc = hidden // n_heads
    for i in range(n_layers):
        a = f"{prefix}.attn_layers.{i}."
        sd[a + "emb_rel_k"] = hash_symmetric(a + "emb_rel_k", (1, 2 * window + 1, kc), math.sqrt(3.0 / kc), seed)
        sd[a + "emb_rel_v"] = hash_symmetric(a + "emb_rel_v", (1, 2 * window + 1, kc), math.sqrt(3.0 / kc), seed)
        for nm in ("conv_q", "conv_k", "conv_v", "conv_o"):
            sd[a + nm + ".weight"] = _w(a + nm + ".weight", (hidden, hidden, 1), hidden, 1.0, seed)
            sd[a + nm + ".bias"] = _b(a + nm + ".bias", hidden, 0.05, seed)
        n1 = f"{prefix}.norm_layers_1.{i}."
        sd[n1 + "gamma"] = 1.0 + hash_symmetric(n1 + "gamma", (hidden,), 0.1, seed)
        sd[n1 + "beta"] = _b(n1 + "beta", hidden, 0.1, seed)
        f = f"{prefix}.ffn_layers.{i}."
        sd[f + "conv_1.weight"] = _w(f + "conv_1.weight", (filt, hidden, ksize), hidden * ksize, 1.0, seed)
        sd[f + "conv_1.bias"] = _b(f + "conv_1.bias", filt, 0.05, seed)
        sd[f + "conv_2.weight"] = _w(f + "conv_2.weight", (hidden, filt, ksize), filt * ksize, 1.0, seed)
        sd[f + "conv_2.bias"] = _b(f + "conv_2.bias", hidden, 0.05, seed)
        n2 = f"{prefix}.norm_layers_2.{i}."
        sd[n2 + "gamma"] = 1.0 + hash_symmetric(n2 + "gamma", (hidden,), 0.1, seed)
        sd[n2 + "beta"] = _b(n2 + "beta", hidden, 0.1, seed)


def make_vits_state_dict(config=None, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    cfg = config or VITS_V2_CONFIG
    m = cfg["model"]
    H = m["hidden_channels"]
    IC = m["inter_channels"]
    FC = m["filter_channels"]
    NH = m["n_heads"]
    NL = m["n_layers"]
    KS = m["kernel_size"]
    GIN = m["gin_channels"]
    NSYM = cfg["n_symbols"]
    # fixed by the reference's constructors, not by hps: ssl_dim 768 and 1024 bins
    # (module/models.py:886-892), MRTE hidden 512 (mrte_model.py:13), ref_enc 704 -> 128
    # (module/models.py:884, module/modules.py:675-677)
    SSL, NBINS, MRTE_H, REF_IN, REF_H = 768, 1024, 512, 704, 128
    sd = OrderedDict()
    # ---- enc_p (TextEncoder, reference module/models.py:154-231)
    sd["enc_p.ssl_proj.weight"] = _w("enc_p.ssl_proj.weight", (H, SSL, 1), SSL, 1.0, seed)
    sd["enc_p.ssl_proj.bias"] = _b("enc_p.ssl_proj.bias", H, 0.05, seed)
    _encoder(sd, "enc_p.encoder_ssl", NL // 2, H, FC, NH, KS, seed)
    _encoder(sd, "enc_p.encoder_text", NL, H, FC, NH, KS, seed)
    sd["enc_p.text_embedding.weight"] = hash_symmetric("enc_p.text_embedding", (NSYM, H), 1.0, seed)
    mr = "enc_p.mrte."
    for nm in ("conv_q", "conv_k", "conv_v", "conv_o"):
        sd[mr + "cross_attention." + nm + ".weight"] = _w(mr + nm + ".weight", (MRTE_H, MRTE_H, 1), MRTE_H, 1.0, seed)
        sd[mr + "cross_attention." + nm + ".bias"] = _b(mr + nm + ".bias", MRTE_H, 0.05, seed)
    sd[mr + "c_pre.weight"] = _w(mr + "c_pre.weight", (MRTE_H, H, 1), H, 1.0, seed)
    sd[mr + "c_pre.bias"] = _b(mr + "c_pre.bias", MRTE_H, 0.05, seed)
    sd[mr + "text_pre.weight"] = _w(mr + "text_pre.weight", (MRTE_H, H, 1), H, 1.0, seed)
    sd[mr + "text_pre.bias"] = _b(mr + "text_pre.bias", MRTE_H, 0.05, seed)
    sd[mr + "c_post.weight"] = _w(mr + "c_post.weight", (H, MRTE_H, 1), MRTE_H, 1.0, seed)
    sd[mr + "c_post.bias"] = _b(mr + "c_post.bias", H, 0.05, seed)
    _encoder(sd, "enc_p.encoder2", NL // 2, H, FC, NH, KS, seed)
    sd["enc_p.proj.weight"] = _w("enc_p.proj.weight", (2 * IC, H, 1), H, 0.5, seed)
    sd["enc_p.proj.bias"] = _b("enc_p.proj.bias", 2 * IC, 0.05, seed)
    # ---- dec (Generator, reference module/models.py:407-450)
    UIC = m["upsample_initial_channel"]
    sd["dec.conv_pre.weight"] = _w("dec.conv_pre.weight", (UIC, IC, 7), IC * 7, 1.0, seed)
    sd["dec.conv_pre.bias"] = _b("dec.conv_pre.bias", UIC, 0.05, seed)
    ch = UIC
    for i, (u, k) in enumerate(zip(m["upsample_rates"], m["upsample_kernel_sizes"])):
        cin, cout = UIC // (2 ** i), UIC // (2 ** (i + 1))
        # transposed conv weight is [C_in, C_out, k]; each output sees ~k/u taps of C_in
        _wn(sd, f"dec.ups.{i}", (cin, cout, k), cin * max(k // u, 1), 1.0, seed, bias=True, bias_n=cout)
        ch = cout
        for j, (rk, rd) in enumerate(zip(m["resblock_kernel_sizes"], m["resblock_dilation_sizes"])):
            r = f"dec.resblocks.{i * len(m['resblock_kernel_sizes']) + j}"
            for c in range(len(rd)):
                _wn(sd, f"{r}.convs1.{c}", (ch, ch, rk), ch * rk, 0.7, seed)
                _wn(sd, f"{r}.convs2.{c}", (ch, ch, rk), ch * rk, 0.7, seed)
    sd["dec.conv_post.weight"] = _w("dec.conv_post.weight", (1, ch, 7), ch * 7, 0.3, seed)
    sd["dec.cond.weight"] = _w("dec.cond.weight", (UIC, GIN, 1), GIN, 1.0, seed)
    sd["dec.cond.bias"] = _b("dec.cond.bias", UIC, 0.05, seed)
    # ---- flow (ResidualCouplingBlock, reference module/models.py:253-295)
    half = IC // 2
    for fi in range(4):
        f = f"flow.flows.{2 * fi}"
        sd[f + ".pre.weight"] = _w(f + ".pre.weight", (H, half, 1), half, 1.0, seed)
        sd[f + ".pre.bias"] = _b(f + ".pre.bias", H, 0.05, seed)
        for li in range(4):
            _wn(sd, f"{f}.enc.in_layers.{li}", (2 * H, H, 5), H * 5, 1.0, seed)
            rs = 2 * H if li < 3 else H
            _wn(sd, f"{f}.enc.res_skip_layers.{li}", (rs, H, 1), H, 1.0, seed)
        _wn(sd, f"{f}.enc.cond_layer", (2 * H * 4, GIN, 1), GIN, 1.0, seed)
        sd[f + ".post.weight"] = _w(f + ".post.weight", (half, H, 1), H, 0.5, seed)
        sd[f + ".post.bias"] = _b(f + ".post.bias", half, 0.05, seed)
    # ---- ref_enc (MelStyleEncoder, reference module/modules.py:672-749)
    sd["ref_enc.spectral.0.fc.weight"] = _w("ref_enc.spectral.0", (REF_H, REF_IN), REF_IN, 1.0, seed)
    sd["ref_enc.spectral.0.fc.bias"] = _b("ref_enc.spectral.0.b", REF_H, 0.05, seed)
    sd["ref_enc.spectral.3.fc.weight"] = _w("ref_enc.spectral.3", (REF_H, REF_H), REF_H, 1.0, seed)
    sd["ref_enc.spectral.3.fc.bias"] = _b("ref_enc.spectral.3.b", REF_H, 0.05, seed)
    for ti in range(2):
        sd[f"ref_enc.temporal.{ti}.conv1.conv.weight"] = _w(f"ref_enc.temporal.{ti}", (2 * REF_H, REF_H, 5), REF_H * 5, 1.0, seed)
        sd[f"ref_enc.temporal.{ti}.conv1.conv.bias"] = _b(f"ref_enc.temporal.{ti}.b", 2 * REF_H, 0.05, seed)
    for nm in ("w_qs", "w_ks", "w_vs", "fc"):
        sd[f"ref_enc.slf_attn.{nm}.weight"] = _w(f"ref_enc.slf_attn.{nm}", (REF_H, REF_H), REF_H, 1.0, seed)
        sd[f"ref_enc.slf_attn.{nm}.bias"] = _b(f"ref_enc.slf_attn.{nm}.b", REF_H, 0.05, seed)
    sd["ref_enc.fc.fc.weight"] = _w("ref_enc.fc.fc", (GIN, REF_H), REF_H, 1.0, seed)
    sd["ref_enc.fc.fc.bias"] = _b("ref_enc.fc.fc.b", GIN, 0.05, seed)
    # ---- top-level ssl_proj + RVQ codebook (reference module/models.py:886-892)
    sd["ssl_proj.weight"] = _w("ssl_proj.weight", (SSL, SSL, 2), SSL * 2, 1.0, seed)
    sd["ssl_proj.bias"] = _b("ssl_proj.bias", SSL, 0.05, seed)
    sd["quantizer.vq.layers.0._codebook.embed"] = hash_symmetric("codebook.embed", (NBINS, SSL), 1.0, seed)
    # EMA bookkeeping buffers present in real checkpoints (reference core_vq.py:106-109); `inited`
    # must be true or the reference re-runs k-means on first use (core_vq.py:113-122)
    sd["quantizer.vq.layers.0._codebook.inited"] = torch.ones(1)
    sd["quantizer.vq.layers.0._codebook.cluster_size"] = torch.ones(NBINS)
    sd["quantizer.vq.layers.0._codebook.embed_avg"] = sd["quantizer.vq.layers.0._codebook.embed"].clone()
    if m.get("version") in ("v2Pro", "v2ProPlus"):
        # speaker-verification conditioning (reference module/models.py:895-899): sv_emb 20480 -> gin, PReLU(gin), ge_to512
        sd["sv_emb.weight"] = _w("sv_emb.weight", (GIN, 20480), 20480, 1.0, seed)
        sd["sv_emb.bias"] = _b("sv_emb.bias", GIN, 0.05, seed)
        sd["ge_to512.weight"] = _w("ge_to512.weight", (512, GIN), GIN, 1.0, seed)
        sd["ge_to512.bias"] = _b("ge_to512.bias", 512, 0.05, seed)
        sd["prelu.weight"] = 0.25 + hash_symmetric("prelu.weight", (GIN,), 0.2, seed)
    return sd


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def make_utterances(n: int, prompt_phones: int = 40, target_phones: int = 40, prompt_tokens: int = 100,
                    n_symbols: int = 732, n_codes: int = 1024, seed: int = 0, ragged: bool = False):
    """Per-utterance synthetic inputs in the shape TTS.to_batch hands to infer_panel
    (reference TTS_infer_pack/TTS.py:899-904): all_phones = prompt + target phones,
    bert = zeros [1024, X] (English path, TextPreprocessor.py:216-220)."""
    prompt_ph = hash_ints(f"prompt_phones", prompt_phones, n_symbols, seed).tolist()
    prompt_sem = torch.from_numpy(hash_ints("prompt_semantic", prompt_tokens, n_codes, seed)).long()
    items = []
    for u in range(n):
        lt = target_phones
        if ragged:
            lt = max(4, target_phones - int(hash_ints(f"ragged{u}", 1, max(target_phones // 2, 1), seed)[0]))
        ph = hash_ints(f"phones{u}", lt, n_symbols, seed).tolist()
        items.append({
            "phones": ph,
            "all_phones": prompt_ph + ph,
            "bert": torch.zeros(1024, len(prompt_ph) + len(ph), dtype=torch.float32),
            "norm_text": "x" * lt,
        })
    return {"prompt_phones": prompt_ph, "prompt_semantic": prompt_sem, "items": items}


def make_refer_spec(frames: int = 200, bins: int = 1025, seed: int = 0) -> torch.Tensor:
    """Reference spectrogram stand-in uniform[0,1) [1, bins, frames]."""
    u = hash_uniform("refer_spec", bins * frames, seed)
    return torch.from_numpy(u.reshape(1, bins, frames).copy())


# --------------------------------------------------------------------------
# v3 / v4 vocoders (H15, H16)
# --------------------------------------------------------------------------
HIFIGAN_V4_CONFIG = {   # reference TTS_infer_pack/TTS.py:631-641
    "kind": "hifigan", "initial_channel": 100, "resblock": "1", "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "upsample_rates": [10, 6, 2, 2, 2],
    "upsample_initial_channel": 512, "upsample_kernel_sizes": [20, 12, 4, 4, 4], "gin_channels": 0, "is_bias": True,
}
BIGVGAN_V2_24K_CONFIG = {   # reference BigVGAN/configs/bigvgan_v2_24khz_100band_256x.json
    "kind": "bigvgan", "num_mels": 100, "resblock": "1", "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "upsample_rates": [4, 4, 2, 2, 2, 2],
    "upsample_kernel_sizes": [8, 8, 4, 4, 4, 4], "upsample_initial_channel": 1536, "activation": "snakebeta",
    "snake_logscale": True, "use_tanh_at_final": False, "use_bias_at_final": False,
}


def small_vocoder_config(kind: str):
    base = dict(HIFIGAN_V4_CONFIG if kind == "hifigan" else BIGVGAN_V2_24K_CONFIG)
    base.update({"upsample_rates": [4, 2, 2], "upsample_kernel_sizes": [8, 4, 4], "upsample_initial_channel": 128})
    return base


def make_vocoder_state_dict(cfg: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Plain (weight-norm removed) state dict, as the reference holds the vocoders at inference
    (TTS.py:615,642 call remove_weight_norm())."""
    big = cfg["kind"] == "bigvgan"
    cin = cfg["num_mels"] if big else cfg["initial_channel"]
    UIC = cfg["upsample_initial_channel"]
    sd = OrderedDict()
    sd["conv_pre.weight"] = _w("voc.conv_pre.weight", (UIC, cin, 7), cin * 7, 1.0, seed)
    sd["conv_pre.bias"] = _b("voc.conv_pre.bias", UIC, 0.05, seed)
    ch = UIC
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        ci, co = UIC // (2 ** i), UIC // (2 ** (i + 1))
        un = f"ups.{i}.0" if big else f"ups.{i}"
        sd[un + ".weight"] = _w(f"voc.{un}.weight", (ci, co, k), ci * max(k // u, 1), 1.0, seed)
        sd[un + ".bias"] = _b(f"voc.{un}.bias", co, 0.05, seed)
        ch = co
        for j, rk in enumerate(cfg["resblock_kernel_sizes"]):
            r = f"resblocks.{i * nk + j}"
            for c in range(3):
                for nm in ("convs1", "convs2"):
                    sd[f"{r}.{nm}.{c}.weight"] = _w(f"voc.{r}.{nm}.{c}.weight", (ch, ch, rk), ch * rk, 0.7, seed)
                    sd[f"{r}.{nm}.{c}.bias"] = _b(f"voc.{r}.{nm}.{c}.bias", ch, 0.05, seed)
            if big:
                for a in range(6):
                    sd[f"{r}.activations.{a}.act.alpha"] = hash_symmetric(f"voc.{r}.act{a}.alpha", (ch,), 0.5, seed)
                    sd[f"{r}.activations.{a}.act.beta"] = hash_symmetric(f"voc.{r}.act{a}.beta", (ch,), 0.5, seed)
    if big:
        sd["activation_post.act.alpha"] = hash_symmetric("voc.post.alpha", (ch,), 0.5, seed)
        sd["activation_post.act.beta"] = hash_symmetric("voc.post.beta", (ch,), 0.5, seed)
        sd["conv_post.weight"] = _w("voc.conv_post.weight", (1, ch, 7), ch * 7, 0.05, seed)
    else:
        sd["conv_post.weight"] = _w("voc.conv_post.weight", (1, ch, 7), ch * 7, 0.3, seed)
        sd["conv_post.bias"] = _b("voc.conv_post.bias", 1, 0.05, seed)
    return sd


# --------------------------------------------------------------------------
# v3 / v4 flow-matching DiT (H14)
# --------------------------------------------------------------------------
DIT_V3_CONFIG = {"dim": 1024, "depth": 22, "heads": 16, "dim_head": 64, "ff_mult": 2, "mel_dim": 100, "text_dim": 512,
                 "conv_layers": 4}   # reference module/models.py:1219-1222


def small_dit_config():
    return {"dim": 128, "depth": 2, "heads": 2, "dim_head": 64, "ff_mult": 2, "mel_dim": 100, "text_dim": 64,
            "conv_layers": 2}


def make_dit_state_dict(cfg: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """DiT state dict (reference f5_tts/model/backbones/dit.py; keys as under `cfm.estimator.`)."""
    D, td, md = cfg["dim"], cfg["text_dim"], cfg["mel_dim"]
    inner = cfg["heads"] * cfg["dim_head"]
    sd = OrderedDict()

    def lin(name, o, i, gain=1.0):
        sd[name + ".weight"] = _w("dit." + name + ".weight", (o, i), i, gain, seed)
        sd[name + ".bias"] = _b("dit." + name + ".bias", o, 0.05, seed)

    for nm in ("time_embed", "d_embed"):
        lin(nm + ".time_mlp.0", D, 256)
        lin(nm + ".time_mlp.2", D, D)
    for i in range(cfg["conv_layers"]):
        p = f"text_embed.text_blocks.{i}."
        sd[p + "dwconv.weight"] = _w("dit." + p + "dwconv.weight", (td, 1, 7), 7, 1.0, seed)
        sd[p + "dwconv.bias"] = _b("dit." + p + "dwconv.bias", td, 0.05, seed)
        sd[p + "norm.weight"] = 1.0 + hash_symmetric("dit." + p + "norm.weight", (td,), 0.1, seed)
        sd[p + "norm.bias"] = _b("dit." + p + "norm.bias", td, 0.1, seed)
        lin(p + "pwconv1", 2 * td, td)
        sd[p + "grn.gamma"] = hash_symmetric("dit." + p + "grn.gamma", (1, 1, 2 * td), 0.3, seed)
        sd[p + "grn.beta"] = hash_symmetric("dit." + p + "grn.beta", (1, 1, 2 * td), 0.1, seed)
        lin(p + "pwconv2", td, 2 * td)
    lin("input_embed.proj", D, 2 * md + td)
    for j in (0, 2):
        n = f"input_embed.conv_pos_embed.conv1d.{j}"
        sd[n + ".weight"] = _w("dit." + n + ".weight", (D, D // 16, 31), (D // 16) * 31, 1.0, seed)
        sd[n + ".bias"] = _b("dit." + n + ".bias", D, 0.05, seed)
    for i in range(cfg["depth"]):
        p = f"transformer_blocks.{i}."
        lin(p + "attn_norm.linear", 6 * D, D, 0.5)
        lin(p + "attn.to_q", inner, D)
        lin(p + "attn.to_k", inner, D)
        lin(p + "attn.to_v", inner, D)
        lin(p + "attn.to_out.0", D, inner)
        lin(p + "ff.ff.0.0", D * cfg["ff_mult"], D)
        lin(p + "ff.ff.2", D, D * cfg["ff_mult"])
    lin("norm_out.linear", 2 * D, D, 0.5)
    lin("proj_out", md, D)
    return sd


def make_vits_v3_state_dict(config=None, seed: int = 0, dit_cfg=None) -> "OrderedDict[str, torch.Tensor]":
    """SynthesizerTrnV3 (reference module/models.py:1128-1226): the v2 enc_p / ref_enc / quantizer keys without flow / dec,
    plus bridge, wns1 (Encoder 512/512/512, k5, 8 WN layers), linear_mel (training only) and, when `dit_cfg` is given,
    `cfm.estimator.*`."""
    cfg = config or VITS_V2_CONFIG
    GIN = cfg["model"]["gin_channels"]
    IC = cfg["model"]["inter_channels"]
    sd = OrderedDict((k, v) for k, v in make_vits_state_dict(cfg, seed).items() if not k.startswith(("dec.", "flow.")))
    W, NL = 512, 8
    sd["bridge.0.weight"] = _w("bridge.0.weight", (W, IC, 1), IC, 1.5, seed)
    sd["bridge.0.bias"] = _b("bridge.0.bias", W, 0.05, seed)
    sd["wns1.pre.weight"] = _w("wns1.pre.weight", (W, W, 1), W, 1.0, seed)
    sd["wns1.pre.bias"] = _b("wns1.pre.bias", W, 0.05, seed)
    for li in range(NL):
        _wn(sd, f"wns1.enc.in_layers.{li}", (2 * W, W, 5), W * 5, 1.0, seed)
        _wn(sd, f"wns1.enc.res_skip_layers.{li}", (2 * W if li < NL - 1 else W, W, 1), W, 1.0, seed)
    _wn(sd, "wns1.enc.cond_layer", (2 * W * NL, GIN, 1), GIN, 1.0, seed)
    sd["wns1.proj.weight"] = _w("wns1.proj.weight", (W, W, 1), W, 1.0, seed)
    sd["wns1.proj.bias"] = _b("wns1.proj.bias", W, 0.05, seed)
    sd["linear_mel.weight"] = _w("linear_mel.weight", (100, W, 1), W, 1.0, seed)
    sd["linear_mel.bias"] = _b("linear_mel.bias", 100, 0.05, seed)
    if dit_cfg is not None:
        for k, v in make_dit_state_dict(dit_cfg, seed).items():
            sd["cfm.estimator." + k] = v
    return sd


# --------------------------------------------------------------------------
# HuBERT-base (reference feature_extractor/cnhubert.py: transformers.HubertModel, HubertConfig defaults)
# --------------------------------------------------------------------------
def make_lora_state_dict(base: dict, rank: int = 4, seed: int = 3) -> "OrderedDict[str, torch.Tensor]":
    """a v3 / v4 LoRA checkpoint's `weight` as peft names it for `cfm` wrapped with target_modules [to_q, to_k, to_v, to_out.0]
    (reference TTS.py:561-567): lora_A [rank, in] / lora_B [out, rank] per targeted Linear of the base state dict, fp16"""
    lw = OrderedDict()
    for k, w in base.items():
        if k.startswith("cfm.") and any(k.endswith(t + ".weight") for t in ("to_q", "to_k", "to_v", "to_out.0")):
            stem = "cfm.base_model.model." + k[len("cfm."):-len(".weight")]
            lw[stem + ".lora_A.default.weight"] = hash_symmetric(stem + "A", (rank, w.shape[1]), 0.2, seed).half()
            lw[stem + ".lora_B.default.weight"] = hash_symmetric(stem + "B", (w.shape[0], rank), 0.2, seed).half()
    return lw


def make_hubert_state_dict(seed: int = 0, layers: int = 12, hidden: int = 768, ffn: int = 3072, conv_dim: int = 512,
                           pos_kernel: int = 128, pos_groups: int = 16) -> "OrderedDict[str, torch.Tensor]":
    """fp32 state dict with transformers.HubertModel's key names (weight-normed positional conv as
    parametrizations.weight.original0 / original1)."""
    sd = OrderedDict()
    kernels = (10, 3, 3, 3, 3, 2, 2)
    for i, k in enumerate(kernels):
        cin = 1 if i == 0 else conv_dim
        # gain > 1 keeps the activations from shrinking through seven GELU layers
        sd[f"feature_extractor.conv_layers.{i}.conv.weight"] = _w(f"hubert.conv{i}", (conv_dim, cin, k), cin * k, 2.0, seed)
    sd["feature_extractor.conv_layers.0.layer_norm.weight"] = 1.0 + hash_symmetric("hubert.gn.w", (conv_dim,), 0.1, seed)
    sd["feature_extractor.conv_layers.0.layer_norm.bias"] = _b("hubert.gn.b", conv_dim, 0.1, seed)
    sd["feature_projection.layer_norm.weight"] = 1.0 + hash_symmetric("hubert.fpln.w", (conv_dim,), 0.1, seed)
    sd["feature_projection.layer_norm.bias"] = _b("hubert.fpln.b", conv_dim, 0.1, seed)
    sd["feature_projection.projection.weight"] = _w("hubert.fp.w", (hidden, conv_dim), conv_dim, 1.0, seed)
    sd["feature_projection.projection.bias"] = _b("hubert.fp.b", hidden, 0.05, seed)
    cg = hidden // pos_groups
    sd["encoder.pos_conv_embed.conv.bias"] = _b("hubert.pos.b", hidden, 0.05, seed)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = 0.5 + hash_symmetric("hubert.pos.g", (1, 1, pos_kernel), 0.2, seed)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = _w("hubert.pos.v", (hidden, cg, pos_kernel), cg * pos_kernel, 1.0, seed)
    sd["encoder.layer_norm.weight"] = 1.0 + hash_symmetric("hubert.encln.w", (hidden,), 0.1, seed)
    sd["encoder.layer_norm.bias"] = _b("hubert.encln.b", hidden, 0.1, seed)
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{nm}.weight"] = _w(p + nm + ".w", (hidden, hidden), hidden, 1.0, seed)
            sd[p + f"attention.{nm}.bias"] = _b(p + nm + ".b", hidden, 0.05, seed)
        sd[p + "layer_norm.weight"] = 1.0 + hash_symmetric(p + "ln1.w", (hidden,), 0.1, seed)
        sd[p + "layer_norm.bias"] = _b(p + "ln1.b", hidden, 0.1, seed)
        sd[p + "feed_forward.intermediate_dense.weight"] = _w(p + "f1.w", (ffn, hidden), hidden, 1.0, seed)
        sd[p + "feed_forward.intermediate_dense.bias"] = _b(p + "f1.b", ffn, 0.05, seed)
        sd[p + "feed_forward.output_dense.weight"] = _w(p + "f2.w", (hidden, ffn), ffn, 1.0, seed)
        sd[p + "feed_forward.output_dense.bias"] = _b(p + "f2.b", hidden, 0.05, seed)
        sd[p + "final_layer_norm.weight"] = 1.0 + hash_symmetric(p + "ln2.w", (hidden,), 0.1, seed)
        sd[p + "final_layer_norm.bias"] = _b(p + "ln2.b", hidden, 0.1, seed)
    return sd


def make_eres2net_state_dict(seed: int = 0, m_channels: int = 64, base_width: int = 24, scale: int = 4, expansion: int = 4,
                             num_blocks=(3, 4, 6, 3)) -> "OrderedDict[str, torch.Tensor]":
    """fp32 state dict with the key names of the reference's ERes2NetV2(baseWidth=24, scale=4, expansion=4) (sv.py:14,
    eres2net/ERes2NetV2.py:155-226); BatchNorm running statistics are non-trivial so that the eval-mode folding is exercised.
    The pooling / embedding head (seg_1) that forward3 never uses is left out."""
    import math
    sd = OrderedDict()

    def bn(prefix, n, gamma=1.0):
        sd[prefix + ".weight"] = gamma * (1.0 + hash_symmetric(prefix + ".w", (n,), 0.1, seed))
        sd[prefix + ".bias"] = _b(prefix + ".b", n, 0.05, seed)
        sd[prefix + ".running_mean"] = _b(prefix + ".m", n, 0.05, seed)
        sd[prefix + ".running_var"] = 1.0 + hash_symmetric(prefix + ".v", (n,), 0.2, seed)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.long)

    def conv(name, cout, cin, k, gain=1.0, bias=False):
        sd[name + ".weight"] = _w(name, (cout, cin, k, k), cin * k * k, gain, seed)
        if bias:
            sd[name + ".bias"] = _b(name + ".bias", cout, 0.05, seed)

    def aff(prefix, ch):
        inter = ch // 4
        conv(prefix + ".local_att.0", inter, 2 * ch, 1, 1.0, True)
        bn(prefix + ".local_att.1", inter)
        conv(prefix + ".local_att.3", ch, inter, 1, 1.0, True)
        bn(prefix + ".local_att.4", ch)

    conv("conv1", m_channels, 1, 3, 1.0)
    bn("bn1", m_channels)
    in_planes = m_channels
    for li, (planes, nb, stride) in enumerate(zip((m_channels, 2 * m_channels, 4 * m_channels, 8 * m_channels), num_blocks, (1, 2, 2, 2)), 1):
        width = int(math.floor(planes * (base_width / 64.0)))
        for bi in range(nb):
            p = f"layer{li}.{bi}"
            st = stride if bi == 0 else 1
            conv(p + ".conv1", width * scale, in_planes, 1)
            bn(p + ".bn1", width * scale)
            for i in range(scale):
                conv(p + f".convs.{i}", width, width, 3)
                bn(p + f".bns.{i}", width)
            if li >= 3:
                for j in range(scale - 1):
                    aff(p + f".fuse_models.{j}", width)
            conv(p + ".conv3", planes * expansion, width * scale, 1, 0.7)
            bn(p + ".bn3", planes * expansion)
            if st != 1 or in_planes != planes * expansion:
                conv(p + ".shortcut.0", planes * expansion, in_planes, 1, 0.8)
                bn(p + ".shortcut.1", planes * expansion)
            in_planes = planes * expansion
    conv("layer3_ds", 8 * m_channels * expansion, 4 * m_channels * expansion, 3, 1.0)
    aff("fuse34", 8 * m_channels * expansion)
    return sd


def make_waveform(n: int, seed: int = 0, sr: int = 16000) -> torch.Tensor:
    """a speech-like test signal in [-0.5, 0.5]: a few amplitude-modulated partials + hash noise, [n] fp32"""
    t = np.arange(n, dtype=np.float64) / sr
    f0 = 110.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / sr
    y = sum(np.sin(k * ph + 0.3 * k) / k for k in range(1, 9)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t))
    y = 0.25 * y / np.abs(y).max() + 0.05 * (hash_uniform("wave_noise", n, seed).astype(np.float64) - 0.5)
    return torch.from_numpy(y.astype(np.float32))


# --------------------------------------------------------------------------
# BERT-large (reference TTS.py:472-482: chinese-roberta-wwm-ext-large, a transformers BertForMaskedLM)
# --------------------------------------------------------------------------
BERT_TEST_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("，。！？、：；你好我是小明今天气不错们一起去公园散步吧的了在有人这中大为上个国和地到以说时要就出会可也") + list("abcxyz0123456789")


def make_bert_state_dict(seed: int = 0, layers: int = 24, hidden: int = 1024, ffn: int = 4096, vocab: int = len(BERT_TEST_VOCAB),
                         positions: int = 512) -> "OrderedDict[str, torch.Tensor]":
    """fp32 state dict with transformers.BertModel's key names."""
    sd = OrderedDict()
    sd["embeddings.word_embeddings.weight"] = hash_symmetric("bert.word", (vocab, hidden), 1.0, seed)
    sd["embeddings.position_embeddings.weight"] = hash_symmetric("bert.pos", (positions, hidden), 0.5, seed)
    sd["embeddings.token_type_embeddings.weight"] = hash_symmetric("bert.type", (2, hidden), 0.5, seed)
    sd["embeddings.LayerNorm.weight"] = 1.0 + hash_symmetric("bert.embln.w", (hidden,), 0.1, seed)
    sd["embeddings.LayerNorm.bias"] = _b("bert.embln.b", hidden, 0.1, seed)
    for i in range(layers):
        p = f"encoder.layer.{i}."
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            sd[p + nm + ".weight"] = _w(p + nm + ".w", (hidden, hidden), hidden, 1.0, seed)
            sd[p + nm + ".bias"] = _b(p + nm + ".b", hidden, 0.05, seed)
        sd[p + "attention.output.LayerNorm.weight"] = 1.0 + hash_symmetric(p + "ln1.w", (hidden,), 0.1, seed)
        sd[p + "attention.output.LayerNorm.bias"] = _b(p + "ln1.b", hidden, 0.1, seed)
        sd[p + "intermediate.dense.weight"] = _w(p + "f1.w", (ffn, hidden), hidden, 1.0, seed)
        sd[p + "intermediate.dense.bias"] = _b(p + "f1.b", ffn, 0.05, seed)
        sd[p + "output.dense.weight"] = _w(p + "f2.w", (hidden, ffn), ffn, 1.0, seed)
        sd[p + "output.dense.bias"] = _b(p + "f2.b", hidden, 0.05, seed)
        sd[p + "output.LayerNorm.weight"] = 1.0 + hash_symmetric(p + "ln2.w", (hidden,), 0.1, seed)
        sd[p + "output.LayerNorm.bias"] = _b(p + "ln2.b", hidden, 0.1, seed)
    return sd
