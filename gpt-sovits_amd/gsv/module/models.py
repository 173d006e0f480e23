"""Host-side mirror of the reference's `SynthesizerTrn` inference interface
(reference GPT_SoVITS/module/models.py:796-1010), backed by the HIP engine.

`decode` and `extract_latent` keep the reference's argument meaning and return shapes.
All arithmetic runs in libgsv_hip.so; there is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Union

import torch

from .. import _lib


class SynthesizerTrn:
    _VERSIONS = ("v1", "v2", "v2Pro", "v2ProPlus")
    _FLAVOR = {"v1": 0, "v2": 0, "v2Pro": 0, "v2ProPlus": 0, "v3": 1, "v4": 2}

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0, use_sdp=True,
                 semantic_frame_rate=None, freeze_quantizer=None, version="v2", device="cuda:0", dtype=torch.float16,
                 n_symbols: Optional[int] = None, **kwargs):
        if version not in self._VERSIONS:
            raise NotImplementedError(f"{type(self).__name__} does not implement SoVITS {version}")
        if str(resblock) != "1":
            raise NotImplementedError("only ResBlock1 generators (reference configs/s2.json)")
        if semantic_frame_rate != "25hz":
            raise NotImplementedError("only the 25hz semantic frame rate (reference configs/s2.json)")
        self.version = version
        self.spec_channels = spec_channels
        self.inter_channels = inter_channels
        self.hidden_channels = hidden_channels
        self.upsample_rates = list(upsample_rates)
        self.semantic_frame_rate = semantic_frame_rate
        self.is_v2pro = version in ("v2Pro", "v2ProPlus")       # reference module/models.py:590, 895
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gsv SynthesizerTrn runs on an MI355X (cuda/HIP device) only; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.dtype = dtype
        if n_symbols is None:
            n_symbols = 322 if version == "v1" else 732     # len(text/symbols.py) / len(text/symbols2.py)
        cfg = _lib.VitsConfig()
        cfg.inter_channels, cfg.hidden_channels, cfg.filter_channels = inter_channels, hidden_channels, filter_channels
        cfg.n_heads, cfg.n_layers, cfg.kernel_size = n_heads, n_layers, kernel_size
        cfg.gin_channels, cfg.n_symbols, cfg.ssl_dim, cfg.n_bins = gin_channels, n_symbols, 768, 1024
        cfg.upsample_initial_channel = upsample_initial_channel
        cfg.n_ups = len(upsample_rates)
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.up_rates[i], cfg.up_kernels[i] = u, k
        cfg.n_resblocks = len(resblock_kernel_sizes)
        for j, (k, ds) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            cfg.rb_kernels[j] = k
            for c, d in enumerate(ds):
                cfg.rb_dilations[j][c] = d
        cfg.ref_bins = spec_channels if version == "v1" else 704
        cfg.flavor = self._FLAVOR[version]
        cfg.v2pro = int(self.is_v2pro)
        with torch.cuda.device(self.device):
            _lib.init(idx)
            h = C.c_void_p()
            _lib.check(_lib.lib().gsv_vits_create(C.byref(cfg), _lib.dtype_code(dtype), C.byref(h)), "gsv_vits_create")
            self._h = h
            self.stream = torch.cuda.Stream(device=self.device)
        self._loaded = False
        self._ref_key = None
        self._ref_hold = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().gsv_vits_destroy(h)
            except Exception:
                pass
            self._h = None

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False):
        """Takes the reference checkpoint's `weight` dict (enc_q.* absent or ignored, as with the
        reference's strict=False load, TTS.py:554).  Weight-norm pairs are folded in the library."""
        if self._loaded:
            raise RuntimeError("weights already loaded; create a new SynthesizerTrn")
        l = _lib.lib()
        with torch.cuda.device(self.device):
            for k, v in state_dict.items():
                if not torch.is_tensor(v) or k.startswith("enc_q."):
                    continue
                t = v.detach().to("cpu", torch.float32).contiguous()
                if t.numel() == 0:
                    continue
                _lib.check(l.gsv_vits_load_tensor(self._h, k.encode(), t.data_ptr(), t.numel()), f"load {k}")
            _lib.check(l.gsv_vits_finalize(self._h), "gsv_vits_finalize")
        self._loaded = True
        return self

    # ---- reference audio -------------------------------------------------------------
    def _set_refer(self, refer: Union[torch.Tensor, Sequence[torch.Tensor]], sv_emb=None):
        refs = list(refer) if isinstance(refer, (list, tuple)) else [refer]
        svs = None
        if self.is_v2pro:
            if sv_emb is None:
                raise ValueError("a v2Pro / v2ProPlus model needs sv_emb (one [1, 20480] embedding per reference, sv.py:11-32)")
            svs = list(sv_emb) if isinstance(sv_emb, (list, tuple)) else [sv_emb]
            if len(svs) != len(refs) or any(v.numel() != 20480 for v in svs):
                raise ValueError("sv_emb: expected one [1, 20480] tensor per reference spectrogram")
        elif sv_emb is not None:
            raise ValueError("sv_emb is only used by v2Pro / v2ProPlus models")
        # The key is (address, shape, version) of the caller's tensors; the tensors themselves are held in `_ref_hold`
        # for as long as the key is live, so the caching allocator cannot hand the same address to a different
        # reference spectrogram (the reference recomputes ge from the content on every decode, models.py:966-975).
        key = tuple((r.data_ptr(), tuple(r.shape), r._version) for r in refs + (svs or []))
        if key == self._ref_key:
            return
        keep = [r.to(self.device, torch.float32).contiguous() for r in refs]
        ptrs = (C.c_void_p * len(keep))(*[r.data_ptr() for r in keep])
        frames = (C.c_int * len(keep))(*[int(r.shape[2]) for r in keep])
        bins = int(keep[0].shape[1])
        keep_sv = [v.reshape(-1).to(self.device, torch.float32).contiguous() for v in svs] if svs is not None else None
        # engine stream waits for the conversions above (they run on torch's current stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        if svs is None:
            _lib.check(_lib.lib().gsv_vits_set_refer(self._h, ptrs, frames, bins, len(keep),
                                                     C.c_void_p(self.stream.cuda_stream)), "gsv_vits_set_refer")
        else:
            sv_ptrs = (C.c_void_p * len(keep_sv))(*[v.data_ptr() for v in keep_sv])
            _lib.check(_lib.lib().gsv_vits_set_refer_sv(self._h, ptrs, frames, bins, sv_ptrs, len(keep),
                                                        C.c_void_p(self.stream.cuda_stream)), "gsv_vits_set_refer_sv")
        self.stream.synchronize()
        self._ref_key = key
        self._ref_hold = refs + (svs or [])

    def invalidate_refer(self):
        """Forget the cached reference-audio terms (called by TTS.set_prompt_cache / set_ref_audio)."""
        self._ref_key = None
        self._ref_hold = None

    @torch.no_grad()
    def decode(self, codes: torch.Tensor, text: torch.Tensor, refer, noise_scale: float = 0.5, speed: float = 1,
               sv_emb=None, noise: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
        """reference models.py:961-1005: codes [1,1,T] int64, text [1,L] int64, refer = tensor or list of
        [1, bins, Tr] spectrograms -> waveform [1, 1, 2T*prod(upsample_rates)].
        `noise` (optional, [inter, 2T]) injects the randn_like draw of models.py:1000 for parity tests."""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if codes.numel() == 0 or text.numel() == 0:
            raise ValueError("decode needs at least one semantic token and one phoneme")
        T = int(codes.shape[-1])
        L = int(text.shape[-1])
        up = math.prod(self.upsample_rates)
        with torch.cuda.device(self.device):
            self._set_refer(refer, sv_emb)
            cd = codes.reshape(-1).to(self.device, torch.int32).contiguous()
            tx = text.reshape(-1).to(self.device, torch.int32).contiguous()
            frames = 2 * T if speed == 1 else int(2 * T / speed) + 1      # models.py:226-228
            nz = None
            if noise is not None:
                nz = noise.reshape(self.inter_channels, frames).to(self.device, torch.float32).contiguous()
            wav = torch.empty(frames * up, dtype=torch.float32, device=self.device)
            # after every conversion above: they are enqueued on torch's current stream, the engine reads on its own
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            _lib.check(_lib.lib().gsv_vits_decode(self._h, cd.data_ptr(), T, tx.data_ptr(), L,
                                                  nz.data_ptr() if nz is not None else None, float(noise_scale),
                                                  float(speed), int(seed) & 0xFFFFFFFFFFFFFFFF, wav.data_ptr(),
                                                  C.c_void_p(self.stream.cuda_stream)), "gsv_vits_decode")
            self.stream.synchronize()
        return wav.to(self.dtype).view(1, 1, -1)

    @torch.no_grad()
    def extract_latent(self, x: torch.Tensor) -> torch.Tensor:
        """reference models.py:1007-1010: HuBERT features [1, 768, T50] -> codes [1, 1, T50 // 2] (int64)."""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        T50 = int(x.shape[-1])
        with torch.cuda.device(self.device):
            xs = x.reshape(768, T50).to(self.device, torch.float32).contiguous()
            n = (T50 - 2) // 2 + 1
            out = torch.empty(n, dtype=torch.int32, device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            _lib.check(_lib.lib().gsv_vits_extract_latent(self._h, xs.data_ptr(), T50, out.data_ptr(),
                                                          C.c_void_p(self.stream.cuda_stream)), "gsv_vits_extract_latent")
            self.stream.synchronize()
        return out.long().view(1, 1, -1)

    # ---- test / bench hooks -------------------------------------------------------------
    def debug_tensor(self, name: str, numel: int) -> torch.Tensor:
        out = torch.empty(numel, dtype=torch.float32, device=self.device)
        n = C.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gsv_vits_debug_tensor(self._h, name.encode(), out.data_ptr(), numel, C.byref(n),
                                                        C.c_void_p(self.stream.cuda_stream)))
            self.stream.synchronize()
        return out[: n.value]

    def last_timing(self):
        a, b = C.c_float(0), C.c_float(0)
        _lib.check(_lib.lib().gsv_vits_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


class SynthesizerTrnV3(SynthesizerTrn):
    """Mirror of the reference's `SynthesizerTrnV3` inference interface (module/models.py:1128-1272) for v3 / v4:
    `decode_encp` (enc_p -> bridge -> nearest x1.875 | x2 -> wns1) in the same HIP engine as v2's enc_p, and `cfm`,
    the flow-matching decoder (`CFM` over `DiT`).  The checkpoint's `cfm.estimator.*` keys go to the DiT engine, the
    rest to the encoder engine."""
    _VERSIONS = ("v3", "v4")

    def __init__(self, *args, version="v3", device="cuda:0", dtype=torch.float16, dit_kwargs: Optional[dict] = None, **kwargs):
        super().__init__(*args, version=version, device=device, dtype=dtype, **kwargs)
        from ..f5_tts.model.backbones.dit import DiT
        dk = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4)     # models.py:1219-1222
        if dit_kwargs:
            dk.update(dit_kwargs)
        self.cfm = CFM(100, DiT(**dk, device=device, dtype=dtype))

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False):
        enc = {k: v for k, v in state_dict.items() if not k.startswith("cfm.")}
        dit = {k: v for k, v in state_dict.items() if k.startswith("cfm.estimator.")}
        super().load_state_dict(enc, strict)
        self.cfm.estimator.load_state_dict(dit)
        return self

    def decode(self, *a, **k):
        raise NotImplementedError("v3/v4 models synthesise through decode_encp + cfm.inference + a vocoder (TTS.py:1431-1494)")

    @torch.no_grad()
    def decode_encp(self, codes: torch.Tensor, text: torch.Tensor, refer, ge=None, speed: float = 1):
        """reference models.py:1243-1267: codes [1,1,T], text [1,L], refer [1,bins,Tr] -> (fea [1,512,F], ge).
        `ge` is the handle of the reference audio whose style vector is resident in the engine: pass the value a
        previous call returned (with the same `refer`) to skip recomputing it, exactly like the reference."""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if codes.numel() == 0 or text.numel() == 0:
            raise ValueError("decode_encp needs at least one semantic token and one phoneme")
        T, L = int(codes.shape[-1]), int(text.shape[-1])
        l = _lib.lib()
        with torch.cuda.device(self.device):
            if ge is None or ge != self._ref_key:
                self._set_refer(refer)
            F_ = l.gsv_vits_encp_frames(self._h, T, float(speed))
            if F_ < 1:
                raise ValueError(f"decode_encp: bad length / speed (T={T}, speed={speed})")
            cd = codes.reshape(-1).to(self.device, torch.int32).contiguous()
            tx = text.reshape(-1).to(self.device, torch.int32).contiguous()
            fea = torch.empty(512, F_, dtype=torch.float32, device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            _lib.check(l.gsv_vits_decode_encp(self._h, cd.data_ptr(), T, tx.data_ptr(), L, float(speed), fea.data_ptr(),
                                              C.c_void_p(self.stream.cuda_stream)), "gsv_vits_decode_encp")
            self.stream.synchronize()
        return fea.to(self.dtype).unsqueeze(0), self._ref_key


class Generator:
    """Host-side mirror of the reference's HiFi-GAN `Generator` used as the v4 mel vocoder
    (reference module/models.py:407-471; constructed in TTS_infer_pack/TTS.py:631-648 with
    initial_channel=100, rates (10,6,2,2,2), kernels (20,12,4,4,4), gin_channels=0, is_bias=True)."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0, is_bias=False, device="cuda:0",
                 dtype=torch.float16):
        from .vocoder import _VocoderEngine
        if str(resblock) != "1":
            raise NotImplementedError("only ResBlock1 generators")
        if gin_channels != 0:
            raise NotImplementedError("the conditioned generator is part of SynthesizerTrn.decode")
        self._e = _VocoderEngine(0, initial_channel, upsample_initial_channel, upsample_rates, upsample_kernel_sizes,
                                 resblock_kernel_sizes, resblock_dilation_sizes, bias_at_final=is_bias, tanh_at_final=True,
                                 snake_logscale=False, device=device, dtype=dtype)

    def load_state_dict(self, sd, strict=True):
        self._e.load_state_dict(sd, strict)
        return self

    def remove_weight_norm(self):   # folded at load
        return self

    def eval(self):
        return self

    def __call__(self, x, g=None):
        if g is not None:
            raise NotImplementedError("conditioning input g")
        return self._e(x)


class CFM:
    """Mirror of the reference's `CFM` (module/models.py:1013-1085): `inference` integrates the flow-matching ODE with
    `n_timesteps` Euler steps over the `DiT` estimator, entirely inside the HIP library (`gsv_cfm_inference`).

    `noise` (not in the reference signature) pins the `torch.randn` draw of models.py:1030 for parity tests; left
    None, the draw happens on the device from `seed`.
    """

    def __init__(self, in_channels, dit):
        self.in_channels = in_channels
        self.estimator = dit
        self.sigma_min = 1e-6

    @torch.no_grad()
    def inference(self, mu, x_lens, prompt, n_timesteps, temperature=1.0, inference_cfg_rate=0, noise=None, seed=0):
        """mu [B, T, text_dim]; x_lens unused (as in the reference); prompt [B, in_channels, Tp] -> [B, in_channels, T]"""
        if inference_cfg_rate > 1e-5:
            raise NotImplementedError("classifier-free guidance: every caller in the reference passes inference_cfg_rate=0 "
                                      "(TTS.py:1351, inference_webui.py:937)")
        dit = self.estimator
        if not dit._loaded:
            raise RuntimeError("DiT.load_state_dict() first")
        if mu.dim() != 3 or mu.shape[2] != dit.text_dim or mu.shape[1] < 1:
            raise ValueError(f"expected mu of shape [B, T>=1, {dit.text_dim}], got {tuple(mu.shape)}")
        B, T = int(mu.shape[0]), int(mu.shape[1])
        if prompt.dim() != 3 or prompt.shape[0] not in (1, B) or prompt.shape[1] != self.in_channels or prompt.shape[2] > T:
            raise ValueError(f"expected prompt of shape [{B} or 1, {self.in_channels}, Tp<={T}], got {tuple(prompt.shape)}")
        if prompt.shape[0] != B:           # one prompt broadcast over the batch (models.py:1036 assigns it into every row)
            prompt = prompt.expand(B, -1, -1)
        Tp = int(prompt.shape[2])
        if noise is not None and tuple(noise.shape) != (B, self.in_channels, T):
            raise ValueError(f"noise must have shape {(B, self.in_channels, T)}")
        dev = dit.device
        with torch.cuda.device(dev):
            m = mu.to(dev, torch.float32).contiguous()
            p = prompt.to(dev, torch.float32).contiguous()
            nz = noise.to(dev, torch.float32).contiguous() if noise is not None else None
            out = torch.empty(B, self.in_channels, T, dtype=torch.float32, device=dev)
            dit.stream.wait_stream(torch.cuda.current_stream(dev))
            _lib.check(_lib.lib().gsv_cfm_inference(dit._h, m.data_ptr(), p.data_ptr() if Tp else None, B, T, Tp, int(n_timesteps),
                                                    nz.data_ptr() if nz is not None else None, float(temperature), int(seed),
                                                    out.data_ptr(), C.c_void_p(dit.stream.cuda_stream)), "gsv_cfm_inference")
            dit.stream.synchronize()
        return out.to(mu.dtype if mu.dtype in (torch.float16, torch.float32) else torch.float32)
