"""HuBERT-base content encoder on the HIP library (reference GPT_SoVITS/feature_extractor/cnhubert.py:22-37 wraps
`transformers.HubertModel`; the pipeline calls `cnhubert.model(wav16k.unsqueeze(0))["last_hidden_state"]` on the RAW 16 kHz
waveform, TTS_infer_pack/TTS.py:806-812 -- the Wav2Vec2 feature extractor's normalisation is not applied on that path).

Architecture (HubertConfig defaults = chinese-hubert-base): 7-layer conv feature extractor (kernels 10,3,3,3,3,2,2, strides
5,2,2,2,2,2,2, no bias, GroupNorm(512, 512) after the first conv, GELU), LayerNorm + Linear 512 -> 768, a weight-normed grouped
positional Conv1d(768, 768, k = 128, groups = 16, padding 64, last frame dropped) + GELU added to the sequence, LayerNorm,
12 post-LN encoder layers (12 heads x 64, FFN 3072 GELU).

Host orchestration in Python over the C ABI's single-op entry points, as north_star prescribes: every GEMM / convolution is
`gsv_op_conv1d` (MFMA implicit GEMM, channels-last), attention is `gsv_op_flash_attn64`, normalisations are
`gsv_op_layernorm` / `gsv_op_channel_norm`, the first conv's strided framing is `gsv_op_frame`.  fp16 activations with fp32
accumulation and fp32 normalisation statistics (the reference runs this model in half precision too, TTS.py:329).  There is
no torch compute fallback.  Runs once per reference audio; its output feeds `SynthesizerTrn.extract_latent`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from .. import _lib

ACT_NONE, ACT_GELU = 0, 7          # csrc/common.h activation codes (GELU = erf form, HubertConfig.hidden_act "gelu")

cnhubert_base_path: Optional[str] = None


class _Out(dict):
    """`model(x)["last_hidden_state"]` and `.last_hidden_state`, like the transformers output object"""
    __getattr__ = dict.__getitem__


class HubertModel:
    conv_kernel = (10, 3, 3, 3, 3, 2, 2)
    conv_stride = (5, 2, 2, 2, 2, 2, 2)

    def __init__(self, device="cuda:0", dtype=torch.float16, hidden=768, layers=12, heads=12, ffn=3072, conv_dim=512,
                 pos_kernel=128, pos_groups=16, eps=1e-5):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gsv HubertModel runs on an MI355X (cuda/HIP device) only; there is no CPU path")
        if dtype != torch.float16:
            raise NotImplementedError("the HuBERT engine computes in float16 (fused attention is an fp16 kernel)")
        if hidden // heads != 64:
            raise NotImplementedError("head dim must be 64")
        self.dtype, self.hidden, self.layers, self.heads, self.ffn, self.conv_dim = dtype, hidden, layers, heads, ffn, conv_dim
        self.pos_kernel, self.pos_groups, self.eps = pos_kernel, pos_groups, eps
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.init(idx)
        self.w: Dict[str, torch.Tensor] = {}
        self._loaded = False

    # ---- weights ---------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """HF `HubertModel.state_dict()` keys (an optional "hubert." prefix is dropped).  Conv weights [Cout, Cin, k] become
        [Cout][k * Cin] (tap-major, cin fastest); q/k/v projections are stacked into one [3 * hidden][hidden] GEMM; the
        positional conv's weight norm (`weight_g` / `weight_v`, or parametrizations original0 / original1, norm over
        dims 0 and 1) is folded."""
        sd = {(k[7:] if k.startswith("hubert.") else k): v.detach().float().cpu() for k, v in sd.items() if torch.is_tensor(v)}
        dev, h = self.device, self.hidden

        def put(name, t, half=True):
            self.w[name] = t.to(dev, torch.float16 if half else torch.float32).contiguous()

        def conv_w(t):          # [Cout, Cin, k] -> [Cout][k * Cin]
            return t.permute(0, 2, 1).reshape(t.shape[0], -1)

        w0 = sd["feature_extractor.conv_layers.0.conv.weight"]                 # [512, 1, 10]
        w0p = torch.zeros(w0.shape[0], 16)
        w0p[:, :10] = w0[:, 0, :]
        put("conv0", w0p)
        put("gn_w", sd["feature_extractor.conv_layers.0.layer_norm.weight"], False)
        put("gn_b", sd["feature_extractor.conv_layers.0.layer_norm.bias"], False)
        for i in range(1, 7):
            put(f"conv{i}", conv_w(sd[f"feature_extractor.conv_layers.{i}.conv.weight"]))
        put("fp_ln_w", sd["feature_projection.layer_norm.weight"], False)
        put("fp_ln_b", sd["feature_projection.layer_norm.bias"], False)
        put("fp_w", sd["feature_projection.projection.weight"])
        put("fp_b", sd["feature_projection.projection.bias"], False)
        pre = "encoder.pos_conv_embed.conv."
        if pre + "weight_g" in sd:
            g, v = sd[pre + "weight_g"], sd[pre + "weight_v"]
        elif pre + "parametrizations.weight.original0" in sd:
            g, v = sd[pre + "parametrizations.weight.original0"], sd[pre + "parametrizations.weight.original1"]
        else:
            g, v = None, sd[pre + "weight"]
        pw = v if g is None else v * (g / v.norm(dim=(0, 1), keepdim=True))      # weight_norm(dim=2)
        put("pos_w", conv_w(pw))                                                   # [768][128 * 48], rows grouped by 48
        put("pos_b", sd[pre + "bias"], False)
        put("enc_ln_w", sd["encoder.layer_norm.weight"], False)
        put("enc_ln_b", sd["encoder.layer_norm.bias"], False)
        for i in range(self.layers):
            p = f"encoder.layers.{i}."
            put(f"l{i}.qkv_w", torch.cat([sd[p + "attention.q_proj.weight"], sd[p + "attention.k_proj.weight"],
                                           sd[p + "attention.v_proj.weight"]], 0))
            put(f"l{i}.qkv_b", torch.cat([sd[p + "attention.q_proj.bias"], sd[p + "attention.k_proj.bias"],
                                           sd[p + "attention.v_proj.bias"]], 0), False)
            put(f"l{i}.o_w", sd[p + "attention.out_proj.weight"])
            put(f"l{i}.o_b", sd[p + "attention.out_proj.bias"], False)
            put(f"l{i}.ln1_w", sd[p + "layer_norm.weight"], False)
            put(f"l{i}.ln1_b", sd[p + "layer_norm.bias"], False)
            put(f"l{i}.f1_w", sd[p + "feed_forward.intermediate_dense.weight"])
            put(f"l{i}.f1_b", sd[p + "feed_forward.intermediate_dense.bias"], False)
            put(f"l{i}.f2_w", sd[p + "feed_forward.output_dense.weight"])
            put(f"l{i}.f2_b", sd[p + "feed_forward.output_dense.bias"], False)
            put(f"l{i}.ln2_w", sd[p + "final_layer_norm.weight"], False)
            put(f"l{i}.ln2_b", sd[p + "final_layer_norm.bias"], False)
        self._loaded = True
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device and torch.device(device).type == "cuda" and torch.device(device).index not in (None, self.device.index):
            raise NotImplementedError("create the engine on its device")
        return self

    # ---- ops -------------------------------------------------------------------------------------------------------
    def _conv(self, st, x, T_in, Cin, w, Cout, taps=1, stride=1, pad=0, bias=None, act=ACT_NONE, res=None, T_out=None, ldx=None,
              groups=1):
        l = _lib.lib()
        if T_out is None:
            T_out = (T_in + 2 * pad - taps) // stride + 1
        y = torch.empty(T_out, Cout * groups, dtype=torch.float16, device=self.device)
        d = _lib.ConvDesc()
        d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.T_in, d.T_out, d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = T_in, T_out, Cin, Cout, taps, stride, 1, pad
        d.post_act, d.scale = act, 1.0
        if ldx:
            d.ldx = ldx
        if groups > 1:
            d.Z, d.xz, d.wz, d.yz, d.bz = groups, Cin, Cout * taps * Cin, Cout, Cout
            d.ldx, d.ldy = Cin * groups, Cout * groups
        _lib.check(l.gsv_op_conv1d(C.byref(d), _lib.GSV_F16, st), "gsv_op_conv1d")
        return y, T_out

    def _ln(self, st, x, rows, Cn, w, b, res=None):
        y = torch.empty(rows, Cn, dtype=torch.float16, device=self.device)
        _lib.check(_lib.lib().gsv_op_layernorm(x.data_ptr(), res.data_ptr() if res is not None else None, w.data_ptr(), b.data_ptr(),
                                               y.data_ptr(), rows, Cn, self.eps, _lib.GSV_F16, st), "gsv_op_layernorm")
        return y

    # ---- forward ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, input_values: torch.Tensor, **kw):
        """input_values [1, n] 16 kHz waveform -> {"last_hidden_state": [1, T, hidden]} (T = HF's frame count)"""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if input_values.dim() != 2 or input_values.shape[0] != 1:
            raise ValueError(f"expected a [1, n] waveform, got {tuple(input_values.shape)}")
        n = int(input_values.shape[1])
        if n < 400:
            raise ValueError(f"{n} samples are fewer than the feature extractor's receptive field (400)")
        l, dev, w = _lib.lib(), self.device, self.w
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            x = input_values[0].to(dev, torch.float32).contiguous()
            # conv 0: Conv1d(1, 512, 10, stride 5) as frames [T][16] (10 taps, zero padded) x W [512][16]
            T = (n - 10) // 5 + 1
            fr = torch.empty(T, 16, dtype=torch.float16, device=dev)
            _lib.check(l.gsv_op_frame(x.data_ptr(), n, 10, 5, 0, 16, T, fr.data_ptr(), _lib.GSV_F16, st), "gsv_op_frame")
            h, _ = self._conv(st, fr, T, 16, w["conv0"], self.conv_dim)
            scratch = torch.empty(128 * self.conv_dim, dtype=torch.float32, device=dev)
            hn = torch.empty_like(h)
            _lib.check(l.gsv_op_channel_norm(h.data_ptr(), T, self.conv_dim, w["gn_w"].data_ptr(), w["gn_b"].data_ptr(), 1e-5,
                                             ACT_GELU, scratch.data_ptr(), hn.data_ptr(), _lib.GSV_F16, st), "gsv_op_channel_norm")
            h = hn
            for i in range(1, 7):
                h, T = self._conv(st, h, T, self.conv_dim, w[f"conv{i}"], self.conv_dim, taps=self.conv_kernel[i],
                                  stride=self.conv_stride[i], act=ACT_GELU)
            # feature projection
            h = self._ln(st, h, T, self.conv_dim, w["fp_ln_w"], w["fp_ln_b"])
            h, _ = self._conv(st, h, T, self.conv_dim, w["fp_w"], self.hidden, bias=w["fp_b"])
            # positional conv (grouped, k = 128, pad 64 gives T + 1 frames; HubertSamePadLayer drops the last) + GELU, added
            g = self.pos_groups
            pc, _ = self._conv(st, h, T, self.hidden // g, w["pos_w"], self.hidden // g, taps=self.pos_kernel, pad=self.pos_kernel // 2,
                               bias=w["pos_b"], act=ACT_GELU, T_out=T, groups=g)
            h = self._ln(st, h, T, self.hidden, w["enc_ln_w"], w["enc_ln_b"], res=pc)
            vt = torch.empty(self.hidden * ((T + 31) // 32 * 32), dtype=torch.float16, device=dev)
            for i in range(self.layers):
                qkv, _ = self._conv(st, h, T, self.hidden, w[f"l{i}.qkv_w"], 3 * self.hidden, bias=w[f"l{i}.qkv_b"])
                att = torch.empty(T, self.hidden, dtype=torch.float16, device=dev)
                _lib.check(l.gsv_op_flash_attn64(qkv.data_ptr(), T, self.heads, 0.125, vt.data_ptr(), att.data_ptr(), st),
                           "gsv_op_flash_attn64")
                o, _ = self._conv(st, att, T, self.hidden, w[f"l{i}.o_w"], self.hidden, bias=w[f"l{i}.o_b"], res=h)
                h = self._ln(st, o, T, self.hidden, w[f"l{i}.ln1_w"], w[f"l{i}.ln1_b"])
                f, _ = self._conv(st, h, T, self.hidden, w[f"l{i}.f1_w"], self.ffn, bias=w[f"l{i}.f1_b"], act=ACT_GELU)
                f2, _ = self._conv(st, f, T, self.ffn, w[f"l{i}.f2_w"], self.hidden, bias=w[f"l{i}.f2_b"], res=h)
                h = self._ln(st, f2, T, self.hidden, w[f"l{i}.ln2_w"], w[f"l{i}.ln2_b"])
        return _Out(last_hidden_state=h.view(1, T, self.hidden))


class CNHubert:
    """reference feature_extractor/cnhubert.py:22-37: `.model` is the HubertModel; weights come from `base_path`
    (`pytorch_model.bin` read with a non-executing loader, or `model.safetensors`)."""

    def __init__(self, base_path: Optional[str] = None, device="cuda:0", dtype=torch.float16, state_dict: Optional[dict] = None):
        self.model = HubertModel(device=device, dtype=dtype)
        if state_dict is None:
            base_path = base_path or cnhubert_base_path
            if base_path is None or not os.path.exists(base_path):
                raise FileNotFoundError(base_path)
            st = os.path.join(base_path, "model.safetensors")
            if os.path.exists(st):
                from safetensors.torch import load_file
                state_dict = load_file(st)
            else:
                state_dict = torch.load(os.path.join(base_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        self.model.load_state_dict(state_dict)

    def eval(self):
        return self

    def half(self):
        return self

    def to(self, device):
        return self

    def forward(self, x):
        return self.model(x)["last_hidden_state"]

    __call__ = forward


def get_model():
    return CNHubert()
