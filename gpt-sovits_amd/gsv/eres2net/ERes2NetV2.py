"""ERes2NetV2 speaker encoder (reference GPT_SoVITS/eres2net/ERes2NetV2.py:155-258, blocks :28-151, AFF eres2net/fusion.py:8-27),
inference only, `forward3` = the [B, 20480] time-averaged fused feature map the v2Pro / v2ProPlus SoVITS models are conditioned on.

Layout.  A feature map [C, F, T] lives as fp32 [T][F + 2][C]: time-major, then frequency with one zero row on each side, channels
fastest.  With that layout the 3 x 3 window of a Conv2d reads 3C *contiguous* values per time step -- rows f-1, f, f+1 of all C
channels -- so a Conv2d(C, C', 3, padding=1, stride=s) is ONE batched call of the library's channels-last conv1d GEMM
(`gsv_op_conv1d`): taps = 3 over time, Cin = 3C, one batch slice per output frequency row (x stride s*C, y stride C'), and the
zero rows supply the frequency padding.  1 x 1 convs are the same call with taps = 1; BatchNorm (eval) is folded into weights and
bias at load time; channel concatenations never materialise (a conv over cat(a, b) is two calls chained through the residual
input); the clipped ReLU, SiLU and tanh are GEMM epilogues.  The only non-GEMM kernels are the AFF mix and the final time mean.
fp32 throughout (exact-f32 MFMA): this runs once per reference audio, not per synthesis step.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from .. import _lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SILU, ACT_RELU20 = 0, 1, 2, 6, 11          # csrc/common.h


class _Map:
    """feature map [T][F + 2][C] fp32 (rows 0 and F + 1 of the frequency axis stay zero)"""

    def __init__(self, T: int, Fq: int, Cn: int, device, zero: bool = True):
        self.T, self.F, self.C = T, Fq, Cn
        self.t = (torch.zeros if zero else torch.empty)(T, Fq + 2, Cn, dtype=torch.float32, device=device)

    @property
    def ld(self):
        return (self.F + 2) * self.C

    def interior(self) -> int:
        return self.t.data_ptr() + 4 * self.C


class ERes2NetV2:
    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", m_channels: int = 64, feat_dim: int = 80,
                 baseWidth: int = 26, scale: int = 2, expansion: int = 2, num_blocks=(3, 4, 6, 3), **unused):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the gsv ERes2NetV2 engine runs on an MI355X (cuda/HIP device) only")
        self.m_channels, self.feat_dim, self.baseWidth, self.scale, self.expansion = m_channels, feat_dim, baseWidth, scale, expansion
        self.plan = []
        in_planes = m_channels
        for li, (planes, nb, stride) in enumerate(zip((m_channels, 2 * m_channels, 4 * m_channels, 8 * m_channels), num_blocks, (1, 2, 2, 2)), 1):
            width = int(math.floor(planes * (baseWidth / 64.0)))
            if width % 4 or (li >= 3 and (width // 4) % 4):
                raise NotImplementedError(f"sub-band width {width} must be a multiple of 4 (16 for the fused stages)")
            for bi in range(nb):
                st = stride if bi == 0 else 1
                self.plan.append(dict(p=f"layer{li}.{bi}", stride=st, width=width, fuse=li >= 3, planes=planes * expansion,
                                      shortcut=st != 1 or in_planes != planes * expansion, cin=in_planes, last3=(li == 3 and bi == nb - 1)))
                in_planes = planes * expansion
        self.w: Dict[str, torch.Tensor] = {}
        self._loaded = False
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _fold(sd, conv: str, bn: Optional[str]):
        """Conv2d weight [co][ci][kf][kt] (+ bias) followed by an eval BatchNorm -> (weight, bias) float64"""
        w = sd[conv + ".weight"].double()
        b = sd[conv + ".bias"].double() if conv + ".bias" in sd else torch.zeros(w.shape[0], dtype=torch.float64)
        if bn is not None:
            g = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
            w = w * g.view(-1, 1, 1, 1)
            b = (b - sd[bn + ".running_mean"].double()) * g + sd[bn + ".bias"].double()
        return w, b

    def _put(self, name, w, b=None):
        """[co][ci][kf][kt] -> tap-major GEMM operand [co][kt][kf * ci_count + ci]"""
        co, ci, kf, kt = w.shape
        self.w[name + ".w"] = w.permute(0, 3, 2, 1).reshape(co, kt * kf * ci).float().contiguous().to(self.device)
        if b is not None:
            self.w[name + ".b"] = b.float().contiguous().to(self.device)

    def load_state_dict(self, sd, strict: bool = True):
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        Fq, Cm = self.feat_dim, self.m_channels
        # stem Conv2d(1, C, 3): a banded dense operator over the padded frequency axis, [F * C][kt][F + 2 -> ld]
        w, b = self._fold(sd, "conv1", "bn1")
        ld = (Fq + 2 + 7) // 8 * 8
        dense = torch.zeros(Fq, Cm, 3, ld, dtype=torch.float64)
        for f in range(Fq):
            for kf in range(3):
                dense[f, :, :, f + kf] = w[:, 0, kf, :]
        self.w["stem.w"] = dense.reshape(Fq * Cm, 3 * ld).float().contiguous().to(self.device)
        self.w["stem.b"] = b.repeat(Fq).float().contiguous().to(self.device)
        self.stem_ld = ld
        for blk in self.plan:
            p = blk["p"]
            w, b = self._fold(sd, p + ".conv1", p + ".bn1")
            self._put(p + ".conv1", w, b)
            for i in range(self.scale):
                w, b = self._fold(sd, p + f".convs.{i}", p + f".bns.{i}")
                self._put(p + f".convs.{i}", w, b)
                if blk["fuse"] and i > 0:
                    self._load_aff(sd, p + f".fuse_models.{i - 1}")
            w, b = self._fold(sd, p + ".conv3", p + ".bn3")
            self._put(p + ".conv3", w, b)
            if blk["shortcut"]:
                w, b = self._fold(sd, p + ".shortcut.0", p + ".shortcut.1")
                self._put(p + ".shortcut", w, b)
        self._put("layer3_ds", *self._fold(sd, "layer3_ds", None))
        self._load_aff(sd, "fuse34")
        self._loaded = True
        return self

    def _load_aff(self, sd, p):
        w, b = self._fold(sd, p + ".local_att.0", p + ".local_att.1")
        self._put(p + ".a", w, b)
        w, b = self._fold(sd, p + ".local_att.3", p + ".local_att.4")
        self._put(p + ".b", w, b)

    def eval(self):
        return self

    def half(self):
        return self          # fp32 engine (module docstring)

    def to(self, device):
        return self

    # ---- ops -------------------------------------------------------------------------------------------------------
    def _gemm(self, st, x_ptr, ldx, T_in, Cin, w, w_off, ldw, Cout, y_ptr, ldy, T_out, taps=1, stride=1, pad=0, bias=None, b_off=0,
              act=ACT_NONE, res_ptr=None, Z=1, xz=0, yz=0):
        d = _lib.ConvDesc()
        d.x, d.w, d.y = x_ptr, w.data_ptr() + 4 * w_off, y_ptr
        d.bias = bias.data_ptr() + 4 * b_off if bias is not None else None
        d.res = res_ptr
        d.T_in, d.T_out, d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = T_in, T_out, Cin, Cout, taps, stride, 1, pad
        d.post_act, d.scale, d.out_f32 = act, 1.0, 1
        d.ldx, d.ldw, d.ldy = ldx, ldw, ldy
        if Z > 1:
            d.Z, d.xz, d.wz, d.yz, d.bz = Z, xz, 0, yz, 0
        _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.GSV_F32, st), "gsv_op_conv1d")

    def _conv3x3(self, st, x: _Map, name: str, Cout: int, stride: int, act: int, out: Optional[_Map] = None, res: bool = False,
                 bias: bool = True) -> _Map:
        """Conv2d(C, Cout, 3, padding=1, stride) (+ folded BN, + activation); res: add what `out` already holds before the activation"""
        T_out, F_out = (x.T - 1) // stride + 1, (x.F - 1) // stride + 1
        if out is None:
            out = _Map(T_out, F_out, Cout, self.device)
        w = self.w[name + ".w"]
        self._gemm(st, x.t.data_ptr(), x.ld, x.T, 3 * x.C, w, 0, 9 * x.C, Cout, out.interior(), out.ld, T_out, taps=3, stride=stride, pad=1,
                   bias=self.w[name + ".b"] if bias else None, act=act, res_ptr=out.interior() if res else None, Z=F_out,
                   xz=stride * x.C, yz=Cout)
        return out

    def _conv1x1(self, st, x: _Map, name: str, Cout: int, out: _Map, stride: int = 1, act: int = ACT_NONE, res_ptr=None, bias: bool = True,
                 cin_off: int = 0, cin: Optional[int] = None, cout_off: int = 0, ldw: Optional[int] = None):
        """Conv2d 1 x 1 over the interior rows; (cin_off, cin) select a slice of the weight's input channels (a conv over a channel
        concatenation is one call per part), (cout_off, Cout) a slice of its output channels"""
        cin = x.C if cin is None else cin
        w = self.w[name + ".w"]
        ldw = w.shape[1] if ldw is None else ldw
        self._gemm(st, x.interior(), x.ld, x.T, cin, w, cout_off * ldw + cin_off, ldw, Cout, out.interior(), out.ld, out.T, taps=1,
                   stride=stride, pad=0, bias=self.w[name + ".b"] if bias else None, b_off=cout_off, act=act, res_ptr=res_ptr, Z=out.F,
                   xz=stride * x.C, yz=out.C)

    def _aff(self, st, p: str, x: _Map, y: _Map) -> _Map:
        """fusion.py:22-27: t = tanh(BN(conv(silu(BN(conv(cat(x, y))))))); x (1 + t) + y (1 - t)"""
        Cn, inter = x.C, x.C // 4
        h = _Map(x.T, x.F, inter, self.device, zero=False)
        t = _Map(x.T, x.F, Cn, self.device)                          # zero rows stay finite for the whole-row mix below
        out = _Map(x.T, x.F, Cn, self.device)
        self._conv1x1(st, x, p + ".a", inter, h, bias=False, cin_off=0, cin=Cn)
        self._conv1x1(st, y, p + ".a", inter, h, act=ACT_SILU, res_ptr=h.interior(), cin_off=Cn, cin=Cn)
        self._conv1x1(st, h, p + ".b", Cn, t, act=ACT_TANH)
        # the mix runs over whole rows, zero frequency rows included: they hold 0 in x, y and t, so 0 comes out
        _lib.check(_lib.lib().gsv_op_aff_mix(x.t.data_ptr(), y.t.data_ptr(), t.t.data_ptr(), x.t.numel(), out.t.data_ptr(), st), "gsv_op_aff_mix")
        return out

    def _block(self, st, x: _Map, blk) -> _Map:
        p, s, wd = blk["p"], blk["stride"], blk["width"]
        T2, F2 = (x.T - 1) // s + 1, (x.F - 1) // s + 1
        spx = []
        for i in range(self.scale):                                  # conv1 + bn1 + clipped ReLU, one map per sub-band
            m = _Map(T2, F2, wd, self.device)
            self._conv1x1(st, x, p + ".conv1", wd, m, stride=s, act=ACT_RELU20, cout_off=i * wd)
            spx.append(m)
        outs, sp = [], None
        for i in range(self.scale):
            name = p + f".convs.{i}"
            if i == 0:
                sp = self._conv3x3(st, spx[0], name, wd, 1, ACT_RELU20)
            elif blk["fuse"]:
                sp = self._conv3x3(st, self._aff(st, p + f".fuse_models.{i - 1}", sp, spx[i]), name, wd, 1, ACT_RELU20)
            else:                                                    # conv(sp + spx[i]) = conv(sp) + conv(spx[i])
                o = self._conv3x3(st, sp, name, wd, 1, ACT_NONE, bias=False)
                sp = self._conv3x3(st, spx[i], name, wd, 1, ACT_RELU20, out=o, res=True)
            outs.append(sp)
        out = _Map(T2, F2, blk["planes"], self.device)
        if blk["shortcut"]:
            self._conv1x1(st, x, p + ".shortcut", blk["planes"], out, stride=s)
            first_res = out.interior()
        else:
            first_res = x.interior()                                 # identity shortcut: same shape as `out`
        for i, sp in enumerate(outs):                                # conv3 over cat(outs) + bn3 + shortcut + clipped ReLU
            last = i == self.scale - 1
            self._conv1x1(st, sp, p + ".conv3", blk["planes"], out, act=ACT_RELU20 if last else ACT_NONE, bias=last,
                          res_ptr=first_res if i == 0 else out.interior(), cin_off=i * wd, cin=wd)
        return out

    # ---- forward ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward3(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, T, feat_dim] fbank features -> [B, 8 * m_channels * expansion * feat_dim / 8] (ERes2NetV2.py:246-258)"""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if x.dim() != 3 or x.shape[2] != self.feat_dim:
            raise ValueError(f"expected [B, T, {self.feat_dim}] features, got {tuple(x.shape)}")
        with torch.cuda.device(self.device):
            _lib.init(self.device.index if self.device.index is not None else torch.cuda.current_device())
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            return torch.stack([self._one(st, f) for f in x.to(self.device, torch.float32)])

    def _one(self, st, feat: torch.Tensor) -> torch.Tensor:
        T, Fq, Cm = int(feat.shape[0]), self.feat_dim, self.m_channels
        xin = torch.zeros(T, self.stem_ld, dtype=torch.float32, device=self.device)
        xin[:, 1:Fq + 1] = feat
        out = _Map(T, Fq, Cm, self.device)
        self._gemm(st, xin.data_ptr(), self.stem_ld, T, self.stem_ld, self.w["stem.w"], 0, 3 * self.stem_ld, Fq * Cm, out.interior(), out.ld, T,
                   taps=3, stride=1, pad=1, bias=self.w["stem.b"], act=ACT_RELU)
        out3 = None
        for blk in self.plan:
            out = self._block(st, out, blk)
            if blk["last3"]:
                out3 = out
        out3_ds = self._conv3x3(st, out3, "layer3_ds", out.C, 2, ACT_NONE)
        if (out3_ds.T, out3_ds.F) != (out.T, out.F):
            raise RuntimeError("layer3_ds and layer4 disagree on the map size")
        fused = self._aff(st, "fuse34", out, out3_ds)
        mean = torch.empty(fused.ld, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().gsv_op_time_mean(fused.t.data_ptr(), fused.T, fused.ld, mean.data_ptr(), st), "gsv_op_time_mean")
        return mean.view(fused.F + 2, fused.C)[1:-1].t().reshape(-1)                # (c, f) order of flatten(1, 2)
