"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the v3/v4 flow-matching mel decoder (H14):
`CFM.inference` (reference module/models.py:1027-1085) over the `DiT` estimator (reference
f5_tts/model/backbones/dit.py:88-194, f5_tts/model/modules.py: TimestepEmbedding :656, SinusPositionEmbedding
:152, ConvNeXtV2Block :241, GRN :225, ConvPositionEmbedding :167, AdaLayerNormZero :275,
AdaLayerNormZero_Final :297, FeedForward :318, AttnProcessor :397, DiTBlock :550).  Never imported by the
product path.  Pinned against the reference classes by oracle/gen_golden_vits.py::gen_cfm, EXCEPT the rotary
embedding, which both sides take from oracle/rope.py (x_transformers is absent: PARITY UNPINNED there).
"""
import math

import torch
import torch.nn.functional as F

from . import rope


def precompute_freqs_cis(dim, end, theta=10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    fr = torch.outer(torch.arange(end), freqs).float()
    return torch.cat([torch.cos(fr), torch.sin(fr)], dim=-1)


def timestep_embedding(sd, prefix, t):
    half = 128
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = 1000 * t.unsqueeze(1) * emb.unsqueeze(0)
    h = torch.cat((emb.sin(), emb.cos()), dim=-1)
    h = F.silu(F.linear(h, sd[prefix + ".time_mlp.0.weight"], sd[prefix + ".time_mlp.0.bias"]))
    return F.linear(h, sd[prefix + ".time_mlp.2.weight"], sd[prefix + ".time_mlp.2.bias"])


def text_embed(sd, cfg, text):
    """text [b, n, text_dim] -> same; sinus pos table + ConvNeXtV2 blocks"""
    b, n, d = text.shape
    table = precompute_freqs_cis(d, 4096)
    x = text + table[torch.clamp(torch.arange(n), max=4095)]
    for i in range(cfg["conv_layers"]):
        p = f"text_embed.text_blocks.{i}."
        r = x
        y = F.conv1d(x.transpose(1, 2), sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=d).transpose(1, 2)
        y = F.layer_norm(y, (d,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        y = F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"]))
        gx = torch.norm(y, p=2, dim=1, keepdim=True)
        nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
        y = sd[p + "grn.gamma"] * (y * nx) + sd[p + "grn.beta"] + y
        x = r + F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    return x


def dit_forward(sd, cfg, x0, cond0, t, d, text0, text_cache=None, dt_cache=None):
    """x0, cond0 [b, mel, n]; text0 [b, text_dim, n]; t, d [b] -> v [b, n, mel], text_embed, dt"""
    dim, heads = cfg["dim"], cfg["heads"]
    x, cond, text = x0.transpose(1, 2), cond0.transpose(1, 2), text0.transpose(1, 2)
    b, n, _ = x.shape
    temb = timestep_embedding(sd, "time_embed", t)
    dt = dt_cache if dt_cache is not None else timestep_embedding(sd, "d_embed", d)
    temb = temb + dt
    te = text_cache if text_cache is not None else text_embed(sd, cfg, text)
    h = F.linear(torch.cat((x, cond, te), dim=-1), sd["input_embed.proj.weight"], sd["input_embed.proj.bias"])
    c = h.transpose(1, 2)
    for j in (0, 2):
        c = F.conv1d(c, sd[f"input_embed.conv_pos_embed.conv1d.{j}.weight"], sd[f"input_embed.conv_pos_embed.conv1d.{j}.bias"],
                     padding=15, groups=16)
        c = c * torch.tanh(F.softplus(c))
    h = c.transpose(1, 2) + h
    freqs, _ = rope.RotaryEmbedding(cfg["dim_head"]).forward_from_seq_len(n)
    hd = dim // heads if False else cfg["dim_head"]
    for i in range(cfg["depth"]):
        p = f"transformer_blocks.{i}."
        emb = F.linear(F.silu(temb), sd[p + "attn_norm.linear.weight"], sd[p + "attn_norm.linear.bias"])
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = torch.chunk(emb, 6, dim=1)
        nrm = F.layer_norm(h, (dim,), None, None, 1e-6) * (1 + sc_a[:, None]) + sh_a[:, None]
        q = F.linear(nrm, sd[p + "attn.to_q.weight"], sd[p + "attn.to_q.bias"])
        k = F.linear(nrm, sd[p + "attn.to_k.weight"], sd[p + "attn.to_k.bias"])
        v = F.linear(nrm, sd[p + "attn.to_v.weight"], sd[p + "attn.to_v.bias"])
        q = rope.apply_rotary_pos_emb(q, freqs, 1.0)
        k = rope.apply_rotary_pos_emb(k, freqs, 1.0)
        qh = q.view(b, n, heads, hd).transpose(1, 2)
        kh = k.view(b, n, heads, hd).transpose(1, 2)
        vh = v.view(b, n, heads, hd).transpose(1, 2)
        a = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ vh
        a = a.transpose(1, 2).reshape(b, n, heads * hd)
        a = F.linear(a, sd[p + "attn.to_out.0.weight"], sd[p + "attn.to_out.0.bias"])
        h = h + g_a.unsqueeze(1) * a
        nrm = F.layer_norm(h, (dim,), None, None, 1e-6) * (1 + sc_m[:, None]) + sh_m[:, None]
        f = F.gelu(F.linear(nrm, sd[p + "ff.ff.0.0.weight"], sd[p + "ff.ff.0.0.bias"]), approximate="tanh")
        f = F.linear(f, sd[p + "ff.ff.2.weight"], sd[p + "ff.ff.2.bias"])
        h = h + g_m.unsqueeze(1) * f
    emb = F.linear(F.silu(temb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)
    h = F.layer_norm(h, (dim,), None, None, 1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"]), te, dt


@torch.no_grad()
def cfm_inference(sd, cfg, mu, prompt, n_timesteps, noise, temperature=1.0):
    """mu [B, T, text_dim]; prompt [B or 1, mel, Tp]; noise [B, mel, T] (the randn draw of models.py:1030)."""
    B, T = mu.shape[0], mu.shape[1]
    x = noise * temperature
    Tp = prompt.shape[-1]
    prompt_x = torch.zeros_like(x)
    prompt_x[..., :Tp] = prompt[..., :Tp]
    x[..., :Tp] = 0
    mu_t = mu.transpose(2, 1)
    t, d = 0.0, 1.0 / n_timesteps
    text_cache = dt_cache = None
    for _ in range(n_timesteps):
        tt = torch.ones(B) * t
        dd = torch.ones(B) * d
        v, te, dt = dit_forward(sd, cfg, x, prompt_x, tt, dd, mu_t, text_cache, dt_cache)
        text_cache, dt_cache = te, dt
        x = x + d * v.transpose(2, 1)
        t = t + d
        x[:, :, :Tp] = 0
    return x
