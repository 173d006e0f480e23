"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's SoVITS v2
waveform decoder `SynthesizerTrn.decode`.  Never imported by the product path.

Pinned against the reference itself by oracle/gen_golden_vits.py (imports
/root/reference's SynthesizerTrn here, same synthetic checkpoint, same injected
noise) -> tests/golden/vits_*.npz.

Plain torch fp32 CPU ops on [C, T] tensors, written from the algorithm:
  * decode orchestration (H7)        reference module/models.py:961-1005
  * codebook gather + x2 (H8)        reference module/core_vq.py:181-183, models.py:989-991
  * MelStyleEncoder ref_enc (H9)     reference module/modules.py:672-749
  * TextEncoder enc_p (H10)          reference module/models.py:212-231, attentions.py:64-84,
                                     227-258, 294-323, 366-374, mrte_model.py:25-44
  * flow reverse (H11)               reference module/models.py:288-295, modules.py:434-453, 182-207
  * HiFi-GAN generator (H12)         reference module/models.py:452-471, modules.py:293-306
  * extract_latent (H6)              reference module/models.py:1007-1010, core_vq.py:172-176
Single sequence => every x_mask in the reference is all-ones and is dropped here.
Weight-norm is folded at load: w = g * v / ||v|| (torch.nn.utils.weight_norm, dim=0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            wv = sd[base + ".weight_v"].float()
            g = v.float()
            norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[base + ".weight"] = wv * (g / norm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v.float()
    return out


class VitsOracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: dict):
        self.sd = fold_weight_norm({k: v.detach().cpu() for k, v in state_dict.items()})
        m = config["model"]
        self.m = m
        self.H = m["hidden_channels"]
        self.n_heads = m["n_heads"]
        self.n_layers = m["n_layers"]
        self.ks = m["kernel_size"]
        self.window = 4
        self.up_rates = m["upsample_rates"]
        self.up_ks = m["upsample_kernel_sizes"]
        self.rb_ks = m["resblock_kernel_sizes"]
        self.rb_ds = m["resblock_dilation_sizes"]
        self.inter = m["inter_channels"]

    # ---- small helpers -----------------------------------------------------------
    def conv(self, x, name, dilation=1, padding=None, stride=1):
        w = self.sd[name + ".weight"]
        b = self.sd.get(name + ".bias")
        k = w.shape[-1]
        if padding is None:
            padding = (k * dilation - dilation) // 2
        return F.conv1d(x.unsqueeze(0), w, b, stride=stride, padding=padding, dilation=dilation)[0]

    def lin(self, x, name):
        return F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"))

    def cln(self, x, name):
        """channel LayerNorm on [C, T] (reference modules.py:29-32)."""
        return F.layer_norm(x.t(), (x.shape[0],), self.sd[name + ".gamma"], self.sd[name + ".beta"], 1e-5).t()

    # ---- H9 -------------------------------------------------------------------
    def ref_enc(self, refer: torch.Tensor) -> torch.Tensor:
        """refer [1, bins, Tr] -> ge [512, 1]; v2 uses the first 704 bins (models.py:970)."""
        x = refer[0, :704].float().t()                                   # [Tr, 704]
        x = self.lin(x, "ref_enc.spectral.0.fc")
        x = x * torch.tanh(F.softplus(x))
        x = self.lin(x, "ref_enc.spectral.3.fc")
        x = x * torch.tanh(F.softplus(x))
        x = x.t()                                                          # [128, Tr]
        for i in range(2):
            y = self.conv(x, f"ref_enc.temporal.{i}.conv1.conv")
            a, b = y.chunk(2, dim=0)
            x = x + a * torch.sigmoid(b)
        x = x.t()                                                          # [Tr, 128]
        nh, dm = 2, x.shape[1]
        dk = dm // nh
        q = self.lin(x, "ref_enc.slf_attn.w_qs").view(-1, nh, dk).transpose(0, 1)
        k = self.lin(x, "ref_enc.slf_attn.w_ks").view(-1, nh, dk).transpose(0, 1)
        v = self.lin(x, "ref_enc.slf_attn.w_vs").view(-1, nh, dk).transpose(0, 1)
        a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(dm), dim=2)   # temperature sqrt(d_model)
        o = (a @ v).transpose(0, 1).reshape(-1, dm)
        x = self.lin(o, "ref_enc.slf_attn.fc") + x
        x = self.lin(x, "ref_enc.fc.fc")                                   # [Tr, 512]
        return x.mean(dim=0).unsqueeze(-1)                                 # [512, 1]

    # ---- H10 ------------------------------------------------------------------
    def _rel_attention(self, x, prefix):
        """window-4 relative-position MHA on [C, T] (attentions.py:227-258)."""
        C, T = x.shape
        nh = self.n_heads
        kc = C // nh
        q = self.conv(x, prefix + ".conv_q").view(nh, kc, T).transpose(1, 2)
        k = self.conv(x, prefix + ".conv_k").view(nh, kc, T).transpose(1, 2)
        v = self.conv(x, prefix + ".conv_v").view(nh, kc, T).transpose(1, 2)
        qs = q / math.sqrt(kc)
        scores = qs @ k.transpose(1, 2)                                    # [nh, T, T]
        rel_k = self.sd[prefix + ".emb_rel_k"][0]                          # [2w+1, kc]
        rel_v = self.sd[prefix + ".emb_rel_v"][0]
        w = self.window
        rl = qs @ rel_k.t()                                                # [nh, T, 2w+1]; col r <-> offset r-w
        idx = torch.arange(T)
        for r in range(2 * w + 1):
            off = r - w
            i = idx[(idx + off >= 0) & (idx + off < T)]
            scores[:, i, i + off] += rl[:, i, r]
        p = torch.softmax(scores, dim=-1)
        out = p @ v
        for r in range(2 * w + 1):
            off = r - w
            i = idx[(idx + off >= 0) & (idx + off < T)]
            out[:, i, :] += p[:, i, i + off].unsqueeze(-1) * rel_v[r]
        out = out.transpose(1, 2).reshape(C, T)
        return self.conv(out, prefix + ".conv_o")

    def _encoder(self, x, prefix, n_layers):
        """attentions.Encoder.forward (attentions.py:64-84)."""
        for i in range(n_layers):
            y = self._rel_attention(x, f"{prefix}.attn_layers.{i}")
            x = self.cln(x + y, f"{prefix}.norm_layers_1.{i}")
            y = self.conv(x, f"{prefix}.ffn_layers.{i}.conv_1")
            y = self.conv(torch.relu(y), f"{prefix}.ffn_layers.{i}.conv_2")
            x = self.cln(x + y, f"{prefix}.norm_layers_2.{i}")
        return x

    def _mrte(self, ssl_enc, text, ge):
        """mrte_model.py:25-44: 4-head cross attention ssl -> text, no relative positions."""
        p = "enc_p.mrte"
        s = self.conv(ssl_enc, p + ".c_pre")
        t = self.conv(text, p + ".text_pre")
        C, T = s.shape
        nh = 4
        kc = C // nh
        q = self.conv(s, p + ".cross_attention.conv_q").view(nh, kc, T).transpose(1, 2)
        k = self.conv(t, p + ".cross_attention.conv_k").view(nh, kc, -1).transpose(1, 2)
        v = self.conv(t, p + ".cross_attention.conv_v").view(nh, kc, -1).transpose(1, 2)
        a = torch.softmax((q / math.sqrt(kc)) @ k.transpose(1, 2), dim=-1)
        o = (a @ v).transpose(1, 2).reshape(C, T)
        x = self.conv(o, p + ".cross_attention.conv_o") + s + ge
        return self.conv(x, p + ".c_post")

    def enc_p(self, quantized, text_ids, ge, speed=1, hidden_only=False):
        y = self.conv(quantized, "enc_p.ssl_proj")
        y = self._encoder(y, "enc_p.encoder_ssl", self.n_layers // 2)
        t = self.sd["enc_p.text_embedding.weight"][text_ids].t()
        t = self._encoder(t, "enc_p.encoder_text", self.n_layers)
        y = self._mrte(y, t, ge)
        y = self._encoder(y, "enc_p.encoder2", self.n_layers // 2)
        if speed != 1:   # models.py:226-228
            y = F.interpolate(y.unsqueeze(0), size=int(y.shape[-1] / speed) + 1, mode="linear")[0]
        if hidden_only:
            return y
        stats = self.conv(y, "enc_p.proj")
        return stats[: self.inter], stats[self.inter:]

    # ---- H14 (v3 / v4) ----------------------------------------------------------
    @torch.no_grad()
    def decode_encp(self, codes: torch.Tensor, text: torch.Tensor, refer: torch.Tensor, speed: float = 1, version: str = "v3"):
        """SynthesizerTrnV3.decode_encp (models.py:1243-1267): enc_p hidden -> bridge (1x1 + LeakyReLU 0.01) -> nearest
        x1.875 (v3) / x2 (v4) -> wns1 = Encoder(512, 512, 512, 5, 1, 8) with its length mask (models.py:340-364)."""
        ge = self.ref_enc(refer)
        T = codes.shape[-1]
        q = self.sd["quantizer.vq.layers.0._codebook.embed"][codes[0, 0].long()].t().repeat_interleave(2, dim=1)
        y = self.enc_p(q, text[0].long(), ge, speed, hidden_only=True)
        fea = F.leaky_relu(self.conv(y, "bridge.0"), 0.01)
        fea = F.interpolate(fea.unsqueeze(0), scale_factor=(1.875 if version == "v3" else 2), mode="nearest")[0]
        per = 3.875 if version == "v3" else 4
        sizee = int(T * per) if speed == 1 else int(T * per / speed) + 1
        mask = (torch.arange(fea.shape[-1]) < sizee).to(fea.dtype).unsqueeze(0)
        W, NL = 512, 8
        x = self.conv(fea, "wns1.pre") * mask
        out = torch.zeros_like(x)
        gc = self.conv(ge, "wns1.enc.cond_layer")
        for i in range(NL):                                               # modules.WN.forward (modules.py:182-207)
            xin = self.conv(x, f"wns1.enc.in_layers.{i}") + gc[i * 2 * W:(i + 1) * 2 * W]
            acts = torch.tanh(xin[:W]) * torch.sigmoid(xin[W:])
            rs = self.conv(acts, f"wns1.enc.res_skip_layers.{i}")
            if i < NL - 1:
                x = (x + rs[:W]) * mask
                out = out + rs[W:]
            else:
                out = out + rs
        out = out * mask
        return (self.conv(out, "wns1.proj") * mask).unsqueeze(0), ge

    # ---- H11 ------------------------------------------------------------------
    def _wn(self, x, g, prefix):
        """WN.forward (modules.py:182-207): 4 gated conv layers, dilation 1, k 5."""
        H = self.H
        out = torch.zeros_like(x)
        gc = self.conv(g, prefix + ".cond_layer")                          # [2H*4, 1]
        for i in range(4):
            xin = self.conv(x, f"{prefix}.in_layers.{i}") + gc[i * 2 * H:(i + 1) * 2 * H]
            acts = torch.tanh(xin[:H]) * torch.sigmoid(xin[H:])
            rs = self.conv(acts, f"{prefix}.res_skip_layers.{i}")
            if i < 3:
                x = x + rs[:H]
                out = out + rs[H:]
            else:
                out = out + rs
        return out

    def flow_reverse(self, z, ge):
        """ResidualCouplingBlock reverse (models.py:288-295): for each of the 4 couplings in
        reverse order: Flip, then x1 -= post(WN(pre(x0)))."""
        half = self.inter // 2
        for fi in reversed(range(4)):
            z = torch.flip(z, [0])
            f = f"flow.flows.{2 * fi}"
            x0, x1 = z[:half], z[half:]
            h = self.conv(x0, f + ".pre")
            h = self._wn(h, ge, f + ".enc")
            m = self.conv(h, f + ".post")
            z = torch.cat([x0, x1 - m], 0)
        return z

    # ---- H12 ------------------------------------------------------------------
    def generator(self, z, ge, collect: Optional[dict] = None):
        x = self.conv(z, "dec.conv_pre") + self.conv(ge, "dec.cond")
        nk = len(self.rb_ks)
        for i, (u, k) in enumerate(zip(self.up_rates, self.up_ks)):
            x = F.leaky_relu(x, 0.1)
            x = F.conv_transpose1d(x.unsqueeze(0), self.sd[f"dec.ups.{i}.weight"], self.sd[f"dec.ups.{i}.bias"],
                                   stride=u, padding=(k - u) // 2)[0]
            xs = None
            for j in range(nk):
                r = f"dec.resblocks.{i * nk + j}"
                xr = x
                for c, d in enumerate(self.rb_ds[j]):
                    xt = self.conv(F.leaky_relu(xr, 0.1), f"{r}.convs1.{c}", dilation=d)
                    xt = self.conv(F.leaky_relu(xt, 0.1), f"{r}.convs2.{c}", dilation=1)
                    xr = xt + xr
                xs = xr if xs is None else xs + xr
            x = xs / nk
            if collect is not None:
                collect[f"stage{i}"] = x
        x = F.leaky_relu(x)          # default slope 0.01 (models.py:467)
        x = self.conv(x, "dec.conv_post")
        return torch.tanh(x)

    # ---- H7 -------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, codes: torch.Tensor, text: torch.Tensor, refer, noise_scale: float = 0.5,
               noise: Optional[torch.Tensor] = None, collect: Optional[dict] = None, speed: float = 1,
               sv_emb=None) -> torch.Tensor:
        """codes [1,1,T] int64, text [1,L] int64, refer: tensor or list of [1,bins,Tr];
        noise: the randn draw of models.py:1000, shape [inter, 2T] (None -> torch RNG).
        Returns [1, 1, 2T*prod(upsample_rates)]."""
        refs = refer if isinstance(refer, (list, tuple)) else [refer]
        ges = []
        for i, r in enumerate(refs):
            g = self.ref_enc(r)                                             # [gin, 1]
            if sv_emb is not None:                                          # v2Pro (models.py:971-975): + sv_emb(sv), PReLU
                sv = sv_emb[i] if isinstance(sv_emb, (list, tuple)) else sv_emb
                g = g + self.lin(sv.float().view(1, -1), "sv_emb").t()
                a = self.sd["prelu.weight"].view(-1, 1)
                g = torch.where(g >= 0, g, a * g)
            ges.append(g)
        ge = torch.stack(ges, 0).mean(0)
        ge_enc = ge if sv_emb is None else self.lin(ge.t(), "ge_to512").t()   # MRTE gets ge_to512(ge) (models.py:997)
        q = self.sd["quantizer.vq.layers.0._codebook.embed"][codes[0, 0].long()].t()   # [768, T]
        q = q.repeat_interleave(2, dim=1)                                   # nearest x2
        m_p, logs_p = self.enc_p(q, text[0].long(), ge_enc, speed)
        if noise is None:
            noise = torch.randn_like(m_p)
        z_p = m_p + noise * torch.exp(logs_p) * noise_scale
        z = self.flow_reverse(z_p, ge)
        if collect is not None:
            collect.update(ge=ge, m_p=m_p, logs_p=logs_p, z_p=z_p, z=z)
        o = self.generator(z, ge, collect)
        return o.unsqueeze(0)

    # ---- H6 -------------------------------------------------------------------
    @torch.no_grad()
    def extract_latent(self, ssl: torch.Tensor) -> torch.Tensor:
        """ssl [1, 768, T50] -> codes [1, 1, T25] (models.py:1007-1010, core_vq.py:172-176)."""
        x = self.conv(ssl[0].float(), "ssl_proj", stride=2, padding=0).t()   # [T25, 768]
        e = self.sd["quantizer.vq.layers.0._codebook.embed"].t()             # [768, 1024]
        dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
        return dist.max(dim=-1).indices.view(1, 1, -1)
