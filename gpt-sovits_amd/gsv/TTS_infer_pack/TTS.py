"""Host-side mirror of the reference pipeline API `TTS_infer_pack.TTS` (reference
GPT_SoVITS/TTS_infer_pack/TTS.py: `TTS_Config`:217, `TTS`:412, `run`:984, `to_batch`:842,
`recovery_order`:957, `audio_postprocess`:1377, `using_vocoder_synthesis`:1431, `..._batched_infer`:1496,
`sola_algorithm`:1611), v1/v2 and v3/v4 paths, on the HIP engines.

What is kept: the `run(inputs) -> generator of (sr, int16 ndarray)` contract with the reference's
keys and defaults, length-bucketed batching, AR -> one time-axis-concatenated `decode` per batch,
per-fragment peak normalisation, fragment silence, original-order recovery, x32768 int16 scaling,
the error protocol (1 s of silence then re-raise) and `stop()`.

Front-ends are separate modules this class wires together: `set_ref_audio(path)` =
WAV -> HuBERT engine -> `extract_latent` (prompt tokens), `spectrogram_torch` (reference spectrogram), for v2Pro the ERes2NetV2
speaker embedding, for v3 / v4 the reference mel; `run({"text": ...})` goes through `TextPreprocessor` (pluggable G2P and
language splitter, BERT engine for zh).  Pre-tokenised `segments` and `set_prompt_cache(...)` remain as the model-level entry.
"""
from __future__ import annotations

import math
import os
import random
import time
import traceback
from typing import Callable, Dict, Generator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..AR.models.t2s_model import Text2SemanticDecoder
from ..module.models import SynthesizerTrn, SynthesizerTrnV3
from ..process_ckpt import load_sovits_new  # noqa: F401  (reference process_ckpt.py:129-138; re-exported)


spec_min, spec_max = -12, 2          # reference TTS.py:55-56


def norm_spec(x):
    return (x - spec_min) / (spec_max - spec_min) * 2 - 1


def denorm_spec(x):
    return (x + 1) / 2 * (spec_max - spec_min) + spec_min


def mel_fn(x):
    """reference TTS.py:67-79 (v3: 24 kHz, n_fft 1024, hop 256, 100 mels)"""
    from ..module.mel_processing import mel_spectrogram_torch
    return mel_spectrogram_torch(x, n_fft=1024, win_size=1024, hop_size=256, num_mels=100, sampling_rate=24000, fmin=0, fmax=None,
                                 center=False)


def mel_fn_v4(x):
    """reference TTS.py:81-93 (v4: 32 kHz, n_fft 1280, hop 320, 100 mels)"""
    from ..module.mel_processing import mel_spectrogram_torch
    return mel_spectrogram_torch(x, n_fft=1280, win_size=1280, hop_size=320, num_mels=100, sampling_rate=32000, fmin=0, fmax=None,
                                 center=False)


class NO_PROMPT_ERROR(Exception):
    pass


def set_seed(seed: int) -> int:
    """reference TTS.py:180-205: -1 -> random seed; seeds python/numpy/torch."""
    seed = int(seed)
    seed = seed if seed != -1 else random.randint(0, 2 ** 32 - 1)
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


class TTS_Config:
    """The reference's TTS_Config surface (TTS.py:217-410): `configs` is a dict (the sections "v1" ... "v2ProPlus" and / or
    "custom", or directly the custom section's keys), a YAML file path, or None (defaults).  Differences, all forced by the
    platform: the default device is the GPU ("cuda:0", half precision on) instead of "cpu" -- there is no CPU path -- and a
    weights path that does not exist is kept as given (the engines take in-memory checkpoints too, `TTS.init_*_weights(state=)`)
    instead of silently falling back to the pretrained-model default.  `max_batch` / `max_seq` size the AR engine's K/V arena."""
    default_configs = {
        v: {"device": "cuda:0", "is_half": True, "version": v, "t2s_weights_path": "GPT_SoVITS/pretrained_models/" + t,
            "vits_weights_path": "GPT_SoVITS/pretrained_models/" + s_,
            "cnhuhbert_base_path": "GPT_SoVITS/pretrained_models/chinese-hubert-base",
            "bert_base_path": "GPT_SoVITS/pretrained_models/chinese-roberta-wwm-ext-large"}
        for v, t, s_ in (("v1", "s1bert25hz-2kh-longer-epoch=68e-step=50232.ckpt", "s2G488k.pth"),
                         ("v2", "gsv-v2final-pretrained/s1bert25hz-5kh-longer-epoch=12-step=369668.ckpt", "gsv-v2final-pretrained/s2G2333k.pth"),
                         ("v3", "s1v3.ckpt", "s2Gv3.pth"), ("v4", "s1v3.ckpt", "gsv-v4-pretrained/s2Gv4.pth"),
                         ("v2Pro", "s1v3.ckpt", "v2Pro/s2Gv2Pro.pth"), ("v2ProPlus", "s1v3.ckpt", "v2Pro/s2Gv2ProPlus.pth"))
    }
    v1_languages = ["auto", "en", "zh", "ja", "all_zh", "all_ja"]
    v2_languages = ["auto", "auto_yue", "en", "zh", "ja", "yue", "ko", "all_zh", "all_ja", "all_yue", "all_ko"]
    _KEYS = ("device", "is_half", "version", "t2s_weights_path", "vits_weights_path", "bert_base_path", "cnhuhbert_base_path")

    def __init__(self, configs=None):
        from copy import deepcopy
        self.configs_path = os.path.join("GPT_SoVITS", "configs", "tts_infer.yaml")
        if configs in ["", None]:
            configs = {}
        if isinstance(configs, str):
            self.configs_path = configs
            configs = self._load_configs(configs)
        if not isinstance(configs, dict):
            raise TypeError("configs must be a dict, a YAML path or None")
        sections = deepcopy(self.default_configs)
        if any(k in configs for k in list(self.default_configs) + ["custom"]):
            sections.update(deepcopy(configs))
            custom = sections.get("custom", sections["v2"])
        else:                                   # the custom section's keys given directly
            if configs.get("version", "v2") not in sections:
                raise NotImplementedError(f"version {configs.get('version')} is not one of v1, v2, v2Pro, v2ProPlus, v3, v4")
            custom = dict(sections[configs.get("version", "v2")], **configs)
        self.default_configs = sections
        self.configs = dict(custom)
        self.device = torch.device(custom.get("device", "cuda:0"))
        self.is_half = bool(custom.get("is_half", True))
        self.version = custom.get("version", "v2")
        if self.version not in ("v1", "v2", "v2Pro", "v2ProPlus", "v3", "v4"):
            raise NotImplementedError(f"version {self.version} is not one of v1, v2, v2Pro, v2ProPlus, v3, v4")
        self.t2s_weights_path = custom.get("t2s_weights_path")
        self.vits_weights_path = custom.get("vits_weights_path")
        self.bert_base_path = custom.get("bert_base_path")
        self.cnhuhbert_base_path = custom.get("cnhuhbert_base_path")
        self.max_batch = int(custom.get("max_batch", 32))
        # K/V arena positions per row: the reference's 1500-step loop (t2s_model.py:694) + a 10 s prompt (250 tokens)
        # + phonemes of prompt and text; 2560 x 32 rows x 24 layers is 4 GB of fp16 K/V
        self.max_seq = int(custom.get("max_seq", 2560))
        self.use_vocoder = self.version in ("v3", "v4")     # TTS.py:519-521
        self.languages = self.v1_languages if self.version == "v1" else self.v2_languages
        self.max_sec = None
        self.hz: int = 50
        self.semantic_frame_rate: str = "25hz"
        self.segment_size: int = 20480
        self.filter_length: int = 2048
        self.sampling_rate: int = 32000
        self.hop_length: int = 640
        self.win_length: int = 2048
        self.n_speakers: int = 300
        self.update_configs()

    def _load_configs(self, configs_path: str) -> dict:
        import yaml
        if not os.path.exists(configs_path):
            self.configs = None
            self.save_configs(configs_path)          # reference :361-366: a missing file is created from the defaults
        with open(configs_path, "r", encoding="utf-8") as f:
            return yaml.load(f, Loader=yaml.SafeLoader) or {}

    def save_configs(self, configs_path: Optional[str] = None) -> None:
        import yaml
        from copy import deepcopy
        configs = deepcopy(self.default_configs)
        if getattr(self, "configs", None) is not None:
            configs["custom"] = self.update_configs()
        configs_path = configs_path or self.configs_path
        d = os.path.dirname(configs_path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(configs_path, "w") as f:
            yaml.dump(configs, f)

    def update_configs(self) -> dict:
        self.config = {"device": str(self.device), "is_half": self.is_half, "version": self.version,
                       "t2s_weights_path": self.t2s_weights_path, "vits_weights_path": self.vits_weights_path,
                       "bert_base_path": self.bert_base_path, "cnhuhbert_base_path": self.cnhuhbert_base_path}
        return self.config

    def update_version(self, version: str) -> None:
        self.version = version
        self.use_vocoder = version in ("v3", "v4")
        self.languages = self.v1_languages if self.version == "v1" else self.v2_languages

    def __str__(self):
        self.configs = self.update_configs()
        string = "TTS Config".center(100, "-") + "\n"
        for k, v in self.configs.items():
            string += f"{str(k).ljust(20)}: {str(v)}\n"
        return string + "-" * 100 + "\n"

    __repr__ = __str__

    def __hash__(self):
        return hash(self.configs_path)

    def __eq__(self, other):
        return isinstance(other, TTS_Config) and self.configs_path == other.configs_path

    @property
    def precision(self):
        return torch.float16 if self.is_half else torch.float32


class TTS:
    def __init__(self, configs=None):
        self.configs = configs if isinstance(configs, TTS_Config) else TTS_Config(configs)
        self.t2s_model: Optional[Text2SemanticDecoder] = None
        self.vits_model: Optional[SynthesizerTrn] = None
        self.text_frontend: Optional[Callable] = None
        self.prompt_cache: dict = {
            "ref_audio_path": None, "prompt_semantic": None, "refer_spec": [], "prompt_text": None,
            "prompt_lang": None, "phones": None, "bert_features": None, "norm_text": None, "aux_ref_audio_paths": [],
        }
        self.stop_flag = False
        self.precision = self.configs.precision
        self._t2s_state = None
        self._vits_state = None
        self.vocoder = None
        self.cnhuhbert_model = None
        self.sv_model = None
        self.bert_model = None
        from .TextPreprocessor import TextPreprocessor
        self.text_preprocessor = TextPreprocessor(bert_fn=None, device="cpu")      # reference TTS.py:431-433
        self.vocoder_configs: dict = {"sr": None, "T_ref": None, "T_chunk": None, "upsample_rate": None, "overlapped_len": None}

    # ---- weights (reference TTS.py:484-603) ------------------------------------------------
    def init_t2s_weights(self, weights_path: Optional[str] = None, state: Optional[dict] = None):
        """`state` = {"weight": state_dict, "config": {...}} as stored in the reference's .ckpt
        (TTS.py:590-594); `weights_path` loads such a file with a non-executing loader."""
        if state is None:
            state = torch.load(weights_path, map_location="cpu", weights_only=True)
        self._t2s_state = state
        config = state["config"]
        self.configs.max_sec = config["data"]["max_sec"]
        self.configs.t2s_weights_path = weights_path
        m = Text2SemanticDecoder(config, device=self.configs.device, dtype=self.precision,
                                 max_batch=self.configs.max_batch, max_seq=self.configs.max_seq)
        m.load_state_dict(state["weight"])
        self.t2s_model = m

    def init_vits_weights(self, weights_path: Optional[str] = None, state: Optional[dict] = None, base_state: Optional[dict] = None):
        """reference TTS.py:484-582.  A v3 / v4 LoRA checkpoint (version code 03 / 04, or a `state` carrying "lora_rank") is merged
        into the base model's weights first (`process_ckpt.merge_lora_v3`); the base comes from `base_state` or from the
        configured pretrained path of that version, and a missing base raises FileExistsError like the reference (:491-493)."""
        from ..process_ckpt import get_sovits_version_from_path_fast, merge_lora_v3
        if state is None:
            state = load_sovits_new(weights_path)
        if_lora = "lora_rank" in state
        if weights_path is not None and not if_lora:
            if_lora = bool(get_sovits_version_from_path_fast(weights_path)[2])
        if if_lora:
            if base_state is None:
                path_sovits = self.configs.default_configs[self.configs.version]["vits_weights_path"]
                if not os.path.exists(path_sovits):
                    raise FileExistsError(f"{path_sovits}: SoVITS {self.configs.version} base model missing, cannot load the LoRA weights")
                base_state = load_sovits_new(path_sovits)
            state = dict(state, weight=merge_lora_v3(base_state["weight"], state["weight"], int(state["lora_rank"])))
        self._vits_state = state
        hps = state["config"]
        d, mcfg = hps["data"], dict(hps["model"])
        mcfg.pop("version", None)
        self.configs.filter_length = d["filter_length"]
        self.configs.segment_size = hps["train"]["segment_size"]
        self.configs.sampling_rate = d["sampling_rate"]
        self.configs.hop_length = d["hop_length"]
        self.configs.win_length = d["win_length"]
        self.configs.n_speakers = d["n_speakers"]
        self.configs.vits_weights_path = weights_path
        extra = {}
        cls = SynthesizerTrn
        if self.configs.use_vocoder:
            cls = SynthesizerTrnV3
            if "dit" in hps:             # synthetic / test checkpoints may carry a smaller DiT than models.py:1219-1222
                extra["dit_kwargs"] = hps["dit"]
        v = cls(d["filter_length"] // 2 + 1, hps["train"]["segment_size"] // d["hop_length"],
                n_speakers=d["n_speakers"], version=self.configs.version, device=self.configs.device,
                dtype=self.precision, n_symbols=hps.get("n_symbols"), **extra, **mcfg)
        v.load_state_dict(state["weight"])
        self.vits_model = v
        if getattr(v, "is_v2pro", False) and self.sv_model is None:
            from .. import sv
            if os.path.exists(sv.sv_path):            # reference TTS.py:487-488 loads it here; without the file: init_sv_model(state_dict=)
                self.init_sv_model()

    def init_vocoder(self, version: Optional[str] = None, state: Optional[dict] = None, weights_path: Optional[str] = None):
        """reference TTS.py:605-660: v3 -> BigVGAN-v2 24 kHz x256, v4 -> the HiFi-GAN `Generator` 48 kHz x480.
        `state` = {"config": vocoder hyper-parameters, "weight": state dict} (or a file holding the state dict)."""
        version = version or self.configs.version
        if state is None:
            state = {"weight": torch.load(weights_path, map_location="cpu", weights_only=True), "config": None}
        cfg = state.get("config")
        if version == "v3":
            from ..BigVGAN.bigvgan import BigVGAN
            from .. import synthetic as S
            h = dict(cfg or S.BIGVGAN_V2_24K_CONFIG)
            self.vocoder = BigVGAN(h, device=self.configs.device, dtype=self.precision)
            self.vocoder.load_state_dict(state["weight"])
            self.vocoder_configs.update(sr=24000, T_ref=468, T_chunk=934, upsample_rate=256, overlapped_len=12)
        elif version == "v4":
            from ..module.models import Generator as HifiGenerator
            from .. import synthetic as S
            h = dict(cfg or S.HIFIGAN_V4_CONFIG)
            self.vocoder = HifiGenerator(initial_channel=100, resblock="1", resblock_kernel_sizes=h["resblock_kernel_sizes"],
                                         resblock_dilation_sizes=h["resblock_dilation_sizes"], upsample_rates=h["upsample_rates"],
                                         upsample_initial_channel=h["upsample_initial_channel"],
                                         upsample_kernel_sizes=h["upsample_kernel_sizes"], gin_channels=0, is_bias=True,
                                         device=self.configs.device, dtype=self.precision)
            self.vocoder.load_state_dict(state["weight"])
            self.vocoder_configs.update(sr=48000, T_ref=500, T_chunk=1000, upsample_rate=480, overlapped_len=12)
        else:
            raise ValueError(f"no vocoder for version {version}")
        up = math.prod(h["upsample_rates"])
        if up != self.vocoder_configs["upsample_rate"]:      # reduced test vocoders
            self.vocoder_configs["upsample_rate"] = up
        for k in ("T_ref", "T_chunk", "overlapped_len", "sr"):
            if cfg and k in cfg:
                self.vocoder_configs[k] = cfg[k]

    # ---- prompt cache (replaces set_ref_audio's HuBERT/STFT front-end, TTS.py:737-819) ---------
    def set_prompt_cache(self, prompt_semantic: torch.Tensor, refer_spec: Sequence[torch.Tensor],
                         phones: Optional[List[int]] = None, bert_features: Optional[torch.Tensor] = None,
                         norm_text: str = "", ref_mel: Optional[torch.Tensor] = None,
                         sv_emb: Optional[Sequence[torch.Tensor]] = None):
        """`ref_mel` (v3/v4 only): the log-mel of the reference audio as `mel_fn` / `mel_fn_v4` return it ([1, 100, Tm],
        TTS.py:67-88, 1453); `set_ref_audio` computes it from the waveform (`mel_fn` / `mel_fn_v4`), this is the model-level entry."""
        self.prompt_cache["ref_mel"] = ref_mel
        # v2Pro / v2ProPlus: one speaker-verification embedding [1, 20480] per reference spectrogram (reference sv.py:11-32,
        # TTS.py:790-800 keeps it beside the spectrogram); `set_ref_audio` computes it with gsv.sv.SV (ERes2NetV2)
        self.prompt_cache["sv_emb"] = list(sv_emb) if sv_emb is not None else None
        self.prompt_cache["prompt_semantic"] = prompt_semantic.to(self.configs.device) if prompt_semantic is not None else None
        # the spectrograms go to the device ONCE here (the engine caches the style vector per reference and keys the
        # cache on these tensors), and a new prompt always drops the engine's cached reference terms
        self.prompt_cache["refer_spec"] = [(s.to(self.configs.device), None) for s in refer_spec]
        if self.vits_model is not None:
            self.vits_model.invalidate_refer()
        self.prompt_cache["phones"] = phones
        self.prompt_cache["bert_features"] = bert_features
        self.prompt_cache["norm_text"] = norm_text

    # ---- device / precision (reference TTS.py:677-735) -----------------------------------------------------------
    def enable_half_precision(self, enable: bool = True, save: bool = True):
        """reference TTS.py:677-713: switch every model between fp16 and fp32.  The HIP engines hold their weights in the
        compute dtype, so the models are rebuilt from the retained checkpoints."""
        if str(self.configs.device) == "cpu" and enable:
            raise RuntimeError("half precision needs the GPU")
        if bool(enable) == self.configs.is_half:
            return
        self.configs.is_half = bool(enable)
        self.precision = self.configs.precision
        if save:
            self.configs.save_configs()
        self._rebuild_models()

    def set_device(self, device, save: bool = True):
        """reference TTS.py:715-735: move every model to `device` (another MI355X: there is no CPU path)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("gsv engines run on an MI355X (cuda/HIP device) only; there is no CPU path")
        self.configs.device = device
        if save:
            self.configs.save_configs()
        self._rebuild_models()

    def _rebuild_models(self):
        if self._t2s_state is not None:
            self.t2s_model = None
            self.init_t2s_weights(self.configs.t2s_weights_path, state=self._t2s_state)
        if self._vits_state is not None:
            self.vits_model = None
            self.init_vits_weights(self.configs.vits_weights_path, state=self._vits_state)
        if getattr(self, "_hubert_state", None) is not None:
            self.init_cnhuhbert_weights(state_dict=self._hubert_state)
        if getattr(self, "_bert_state", None) is not None:
            self.init_bert_weights(state_dict=self._bert_state[0], vocab=self._bert_state[1])
        for k in ("prompt_semantic",):
            if self.prompt_cache.get(k) is not None:
                self.prompt_cache[k] = self.prompt_cache[k].to(self.configs.device)
        self.prompt_cache["refer_spec"] = [(sp.to(self.configs.device), a) for sp, a in self.prompt_cache["refer_spec"]]

    def init_bert_weights(self, base_path: Optional[str] = None, state_dict: Optional[dict] = None, vocab=None):
        """chinese-roberta-wwm-ext-large for zh text (reference TTS.py:472-482, TextPreprocessor.py:191-204): hidden_states[-3]
        per character.  `state_dict` / `vocab` (list of tokens) replace the directory for tests."""
        from ..feature_extractor.bert import BertFeature
        self.bert_model = BertFeature(base_path, device=self.configs.device, state_dict=state_dict, vocab=vocab)
        self._bert_state = (state_dict, vocab) if state_dict is not None else None
        self.text_preprocessor.bert_fn = self.bert_model
        self.configs.bert_base_path = base_path

    # ---- reference-audio front-end (reference TTS.py:462-482, 737-819) -------------------------------------------
    def init_cnhuhbert_weights(self, base_path: Optional[str] = None, state_dict: Optional[dict] = None):
        """HuBERT-base content encoder (reference TTS.py:462-470; feature_extractor/cnhubert.py).  `state_dict` = a
        transformers.HubertModel state dict (tests, synthetic weights); `base_path` = directory with pytorch_model.bin /
        model.safetensors."""
        from ..feature_extractor.cnhubert import CNHubert
        self.cnhuhbert_model = CNHubert(base_path, device=self.configs.device, dtype=torch.float16, state_dict=state_dict)
        self._hubert_state = state_dict
        self.configs.cnhuhbert_base_path = base_path

    def set_ref_audio(self, ref_audio_path: str):
        """reference TTS.py:737-747: prompt semantic tokens (HuBERT -> ssl_proj -> VQ) and the reference spectrogram."""
        self._set_prompt_semantic(ref_audio_path)
        self._set_ref_spec(ref_audio_path)
        self.prompt_cache["ref_audio_path"] = ref_audio_path

    def _set_ref_spec(self, ref_audio_path: str):
        spec_audio = self._get_ref_spec(ref_audio_path)
        if self.prompt_cache["refer_spec"] in [[], None]:
            self.prompt_cache["refer_spec"] = [spec_audio]
        else:
            self.prompt_cache["refer_spec"][0] = spec_audio
        if spec_audio[1] is not None:
            # the reference recomputes the embedding inside every run() (TTS.py:1233-1238); it only depends on the reference audio
            emb = self.sv_model.compute_embedding3(spec_audio[1])
            cur = self.prompt_cache.get("sv_emb")
            if cur:
                cur[0] = emb
            else:
                self.prompt_cache["sv_emb"] = [emb]
        if self.vits_model is not None:
            self.vits_model.invalidate_refer()

    def init_sv_model(self, state_dict=None, path: Optional[str] = None):
        """reference TTS.py:672-675 / sv.py:11-23: the ERes2NetV2 speaker-verification model of v2Pro / v2ProPlus"""
        from .. import sv
        if getattr(self, "sv_model", None) is not None and state_dict is None and path is None:
            return
        self.sv_model = sv.SV(self.configs.device, self.configs.is_half, state_dict=state_dict, path=path or sv.sv_path)

    def _get_ref_spec(self, ref_audio_path: str):
        """reference TTS.py:761-800: mono, resampled to the model rate, divided by min(2, peak) when the peak exceeds 1,
        spectrogram_torch(filter_length, hop_length, win_length, center=False)."""
        from ..audio_io import load_wav, resample
        from ..module.mel_processing import spectrogram_torch
        is_v2pro = getattr(self.vits_model, "is_v2pro", False) or self.configs.version in ("v2Pro", "v2ProPlus")
        if is_v2pro and getattr(self, "sv_model", None) is None:
            raise RuntimeError("v2Pro / v2ProPlus: init_sv_model() first (ERes2NetV2 speaker embedding, sv.py:11-32)")
        raw, raw_sr = load_wav(ref_audio_path)
        self.prompt_cache["raw_audio"] = raw                     # [channels, n] float32 at raw_sr (TTS.py:759-762), host side
        self.prompt_cache["raw_sr"] = raw_sr
        self.prompt_cache["ref_mel"] = None                      # v3 / v4: recomputed from raw_audio on first use
        if raw.shape[0] == 2:
            raw = raw.mean(0, keepdims=True)
        audio = torch.from_numpy(resample(raw[:1], raw_sr, self.configs.sampling_rate)).to(self.configs.device)
        maxx = float(audio.abs().max())
        if maxx > 1:
            audio = audio / min(2.0, maxx)
        spec = spectrogram_torch(audio, self.configs.filter_length, self.configs.sampling_rate, self.configs.hop_length,
                                 self.configs.win_length, center=False)
        if self.configs.is_half:
            spec = spec.half()
        audio16k = None
        if is_v2pro:                                             # TTS.py:790-793: the normalised audio again at 16 kHz for the SV model
            audio16k = torch.from_numpy(resample(audio.cpu().numpy(), self.configs.sampling_rate, 16000)).to(self.configs.device)
            if self.configs.is_half:
                audio16k = audio16k.half()
        return spec, audio16k

    def _set_prompt_semantic(self, ref_wav_path: str):
        """reference TTS.py:802-819: 16 kHz audio (3..10 s or OSError) + 0.3 s of silence -> HuBERT last_hidden_state ->
        SynthesizerTrn.extract_latent codes."""
        from ..audio_io import load_wav, resample
        if getattr(self, "cnhuhbert_model", None) is None:
            raise RuntimeError("init_cnhuhbert_weights() first")
        if self.vits_model is None:
            raise RuntimeError("init_vits_weights() first")
        raw, raw_sr = load_wav(ref_wav_path)
        wav16k = resample(raw.mean(0), raw_sr, 16000)
        if wav16k.shape[0] > 160000 or wav16k.shape[0] < 48000:
            raise OSError("参考音频在3~10秒范围外，请更换！")
        zero_wav = np.zeros(int(self.configs.sampling_rate * 0.3), dtype=np.float32)
        wav = torch.from_numpy(np.concatenate([wav16k, zero_wav])).to(self.configs.device)
        hubert_feature = self.cnhuhbert_model.model(wav.unsqueeze(0))["last_hidden_state"].transpose(1, 2)
        codes = self.vits_model.extract_latent(hubert_feature)
        self.prompt_cache["prompt_semantic"] = codes[0, 0].to(self.configs.device)

    def stop(self):
        self.stop_flag = True

    # ---- batching (reference TTS.py:842-973) ------------------------------------------------
    def to_batch(self, data: list, prompt_data: Optional[dict] = None, batch_size: int = 5, threshold: float = 0.75,
                 split_bucket: bool = True, device=torch.device("cpu"), precision=torch.float32):
        """Length-sorted bucketing: a candidate bucket [pos, pos_end) is accepted when its median /
        mean text length ratio reaches `threshold` (or it is a single item), otherwise it shrinks from
        the long end (TTS.py:859-879).  Items then get prompt phones / bert prepended (:899-904)."""
        lens = [len(it["norm_text"]) for it in data]
        batch_index_list: List[List[int]] = []
        if split_bucket:
            order = sorted(range(len(data)), key=lambda i: lens[i])       # stable, like list.sort
            sl = np.array([lens[i] for i in order], dtype=np.float32)
            pos = 0
            while pos < len(order):
                end = min(pos + batch_size, len(order))
                while True:
                    seg = sl[pos:end]
                    score = seg[(end - pos) // 2] / (seg.mean() + 1e-8)
                    if score >= threshold or end - pos == 1:
                        break
                    end -= 1
                batch_index_list.append([order[i] for i in range(pos, end)])
                pos = end
            assert sum(len(b) for b in batch_index_list) == len(data)
        else:
            for i in range(len(data)):
                if i % batch_size == 0:
                    batch_index_list.append([])
                batch_index_list[-1].append(i)
        batches = []
        zero_cache: Dict[int, bool] = {}

        def is_zero(t: Optional[torch.Tensor]) -> bool:
            # all-zero BERT features (every non-zh segment, TextPreprocessor.py:216-220) are passed to the
            # engine as None: bert_proj(0) is its bias, so the [1024, X] tensor never has to reach HBM
            if t is None:
                return True
            k = id(t)
            if k not in zero_cache:
                # host tensors: numpy is ~10x cheaper than a torch dispatch per segment, and an integer max over the bit patterns
                # 3x cheaper than any() (5.5 vs 16.6 us per 330 KB block; -0.0 counts as non-zero, which only forgoes the shortcut)
                if t.device.type == "cpu" and not t.requires_grad and t.dtype == torch.float32 and t.is_contiguous():
                    zero_cache[k] = t.numel() == 0 or int(t.numpy().view(np.uint32).max()) == 0
                elif t.device.type == "cpu" and not t.requires_grad:
                    zero_cache[k] = not t.numpy().any()
                else:
                    zero_cache[k] = not bool(torch.any(t))
            return zero_cache[k]

        for index_list in batch_index_list:
            phones_list, phones_len, all_phones, all_len, all_bert, texts = [], [], [], [], [], []
            max_len = 0
            for idx in index_list:
                it = data[idx]
                ph = torch.LongTensor(it["phones"]).to(device)
                if prompt_data is not None:
                    ap = torch.LongTensor(list(prompt_data["phones"]) + list(it["phones"])).to(device)
                    nb = (prompt_data["bert_features"].shape[-1] if prompt_data["bert_features"] is not None
                          else len(prompt_data["phones"])) + \
                         (it["bert_features"].shape[-1] if it["bert_features"] is not None else len(it["phones"]))
                    if is_zero(prompt_data["bert_features"]) and is_zero(it["bert_features"]):
                        ab = None
                    else:
                        pb = prompt_data["bert_features"]
                        pb = torch.zeros(1024, len(prompt_data["phones"])) if pb is None else pb
                        ib = it["bert_features"]
                        ib = torch.zeros(1024, len(it["phones"])) if ib is None else ib
                        ab = torch.cat([pb, ib], 1).to(dtype=precision, device=device)
                else:
                    ap = ph
                    nb = it["bert_features"].shape[-1] if it["bert_features"] is not None else len(it["phones"])
                    ab = None if is_zero(it["bert_features"]) else it["bert_features"].to(dtype=precision, device=device)
                max_len = max(max_len, nb, ap.shape[-1])
                phones_list.append(ph)
                phones_len.append(ph.shape[-1])
                all_phones.append(ap)
                all_len.append(ap.shape[-1])
                all_bert.append(ab)
                texts.append(it["norm_text"])
            batches.append({
                "phones": phones_list, "phones_len": torch.LongTensor(phones_len).to(device),
                "all_phones": all_phones, "all_phones_len": torch.LongTensor(all_len).to(device),
                "all_bert_features": all_bert, "norm_text": texts, "max_len": max_len,
            })
        return batches, batch_index_list

    def recovery_order(self, data: list, batch_index_list: list) -> list:
        """put batch-ordered fragments back in submission order (TTS.py:957-973)."""
        n = sum(len(b) for b in batch_index_list)
        out = [None] * n
        for i, index_list in enumerate(batch_index_list):
            for j, index in enumerate(index_list):
                out[index] = data[i][j]
        return out

    def audio_postprocess(self, audio: List[List[torch.Tensor]], sr: int, batch_index_list: Optional[list] = None,
                          speed_factor: float = 1.0, split_bucket: bool = True, fragment_interval: float = 0.3,
                          super_sampling: bool = False) -> Tuple[int, np.ndarray]:
        """TTS.py:1377-1429: per fragment divide by its peak if the peak exceeds 1, append
        int(sr*interval) zeros, restore order, concatenate, scale by 32768 and truncate to int16."""
        if super_sampling:
            raise NotImplementedError("audio super-sampling (v3 only) is out of scope")
        import ctypes as C
        from .. import _lib
        dev = torch.device(self.configs.device)
        if dev.type != "cuda":
            raise RuntimeError("audio_postprocess is a HIP kernel (gsv_postprocess); there is no CPU path")
        gap = int(self.configs.sampling_rate * fragment_interval)
        flat = self.recovery_order(audio, batch_index_list) if split_bucket else [f for b in audio for f in b]
        flat = [f.to(dev, self.precision).contiguous().view(-1) for f in flat]
        lens = [int(f.shape[0]) for f in flat]
        self.last_fragment_lengths = [n + gap for n in lens]                                   # used by gsv.sharding
        # one launch (`gsv_postprocess`, csrc/sola.hip): peak, division, gaps, order and the int16 conversion -- the
        # reference loops over fragments on the host with a sync each (TTS.py:1391-1396); only int16 crosses PCIe
        with torch.cuda.device(dev):
            pcm = torch.empty(sum(lens) + gap * len(lens), dtype=torch.int16, device=dev)
            ptrs = (C.c_void_p * max(len(flat), 1))(*[f.data_ptr() for f in flat])
            arr = (C.c_int * max(len(flat), 1))(*lens)
            st = torch.cuda.current_stream(dev)
            _lib.check(_lib.lib().gsv_postprocess(ptrs, arr, len(flat), _lib.GSV_F16 if self.precision == torch.float16 else
                                                  _lib.GSV_F32, gap, pcm.data_ptr(), C.c_void_p(st.cuda_stream)), "gsv_postprocess")
        return sr, self._to_host(pcm)

    def _to_host(self, t: torch.Tensor) -> np.ndarray:
        from gsv.hostcopy import to_host
        return to_host(t)


    # ---- v3 / v4 synthesis (reference TTS.py:1431-1637) -----------------------------------------
    def _prompt_features(self):
        pc = self.prompt_cache
        if pc.get("ref_mel") is None and pc.get("raw_audio") is not None:
            pc["ref_mel"] = self._ref_mel_from_audio(pc["raw_audio"], pc["raw_sr"])
        if pc.get("ref_mel") is None or pc["phones"] is None:
            raise NO_PROMPT_ERROR("v3/v4 need set_ref_audio(path) or set_prompt_cache(..., phones=..., ref_mel=...)")
        dev = self.configs.device
        spec = pc["refer_spec"][0]
        spec = spec[0] if isinstance(spec, tuple) else spec
        spec = spec.to(dev)
        fea_ref, ge = self.vits_model.decode_encp(pc["prompt_semantic"].view(1, 1, -1), torch.as_tensor(pc["phones"]).view(1, -1), spec)
        mel2 = norm_spec(pc["ref_mel"].to(dev, torch.float32))
        T_min = min(mel2.shape[2], fea_ref.shape[2])
        mel2, fea_ref = mel2[:, :, :T_min], fea_ref[:, :, :T_min]
        T_ref = self.vocoder_configs["T_ref"]
        if T_min > T_ref:
            mel2, fea_ref, T_min = mel2[:, :, -T_ref:], fea_ref[:, :, -T_ref:], T_ref
        return spec, fea_ref, ge, mel2.to(self.precision), T_min

    def _ref_mel_from_audio(self, raw_audio: np.ndarray, raw_sr: int) -> torch.Tensor:
        """TTS.py:1442-1453 / 1512-1523: mono mix of a stereo file, resampled to the vocoder's rate (24 kHz v3, 32 kHz v4),
        `mel_fn` / `mel_fn_v4` -> [1, 100, frames] (un-normalised: `_prompt_features` applies norm_spec)."""
        from ..audio_io import resample
        audio = np.asarray(raw_audio, dtype=np.float32)
        if audio.ndim == 1:
            audio = audio[None]
        if audio.shape[0] == 2:
            audio = audio.mean(0, keepdims=True)
        tgt_sr = 24000 if self.configs.version == "v3" else 32000
        audio = torch.from_numpy(resample(audio[:1], raw_sr, tgt_sr)).to(self.configs.device)
        return (mel_fn if self.configs.version == "v3" else mel_fn_v4)(audio)

    @torch.no_grad()
    def using_vocoder_synthesis(self, semantic_tokens: torch.Tensor, phones: torch.Tensor, speed: float = 1.0,
                                sample_steps: int = 32, seed: int = 0, noise_fn: Optional[Callable] = None) -> torch.Tensor:
        """TTS.py:1431-1494: one fragment; the mel is generated chunk by chunk, each chunk prompted with the tail of the
        previous one.  `noise_fn(call_index, shape)` (tests) pins the randn draw of each cfm.inference call."""
        spec, fea_ref, ge, mel2, T_min = self._prompt_features()
        chunk_len = self.vocoder_configs["T_chunk"] - T_min
        fea_todo, ge = self.vits_model.decode_encp(semantic_tokens, phones, spec, ge, speed)
        outs, pos, call = [], 0, 0
        while True:
            chunk = fea_todo[:, :, pos:pos + chunk_len]
            if chunk.shape[-1] == 0:
                break
            pos += chunk_len
            fea = torch.cat([fea_ref, chunk], 2).transpose(2, 1)
            nz = noise_fn(call, (1, 100, fea.shape[1])) if noise_fn else None
            res = self.vits_model.cfm.inference(fea, None, mel2, sample_steps, inference_cfg_rate=0, noise=nz, seed=seed + call)
            res = res[:, :, mel2.shape[2]:]
            call += 1
            mel2 = res[:, :, -T_min:]
            fea_ref = chunk[:, :, -T_min:]
            outs.append(res)
        return self.vocoder(denorm_spec(torch.cat(outs, 2)))[0][0]

    @torch.no_grad()
    def using_vocoder_synthesis_batched_infer(self, idx_list: List[int], semantic_tokens_list: List[torch.Tensor],
                                              batch_phones: List[torch.Tensor], speed: float = 1.0, sample_steps: int = 32,
                                              seed: int = 0, noise_fn: Optional[Callable] = None) -> List[torch.Tensor]:
        """TTS.py:1496-1609: all fragments of a batch concatenated, cut into overlapping chunks that go through ONE
        batched cfm.inference, vocoded as one sequence, re-joined with SOLA and split back per fragment."""
        spec, fea_ref, ge, mel2, T_min = self._prompt_features()
        vc = self.vocoder_configs
        chunk_len = vc["T_chunk"] - T_min
        ov, up = vc["overlapped_len"], vc["upsample_rate"]
        feats, lens = [], []
        for i, idx in enumerate(idx_list):
            f, _ = self.vits_model.decode_encp(semantic_tokens_list[i][-idx:].view(1, 1, -1), batch_phones[i].view(1, -1), spec, ge, speed)
            feats.append(f)
            lens.append(int(f.shape[2]))
        padded = F.pad(torch.cat(feats, 2), (ov, 0))
        chunks, pos, pad_len = [], 0, 0
        while True:
            if pos != 0:
                pos -= ov
            chunk = padded[:, :, pos:pos + chunk_len]
            pos += chunk_len
            if chunk.shape[-1] == 0:
                break
            pad_len = chunk_len - chunk.shape[2]
            if pad_len:
                chunk = F.pad(chunk, (0, pad_len))
            chunks.append(chunk)
        chunks = torch.cat(chunks, 0)
        bs = chunks.shape[0]
        fea = torch.cat([fea_ref.repeat(bs, 1, 1), chunks], 2).transpose(2, 1)
        nz = noise_fn(0, (bs, 100, fea.shape[1])) if noise_fn else None
        pred = self.vits_model.cfm.inference(fea, None, mel2, sample_steps, inference_cfg_rate=0, noise=nz, seed=seed)
        pred = pred[:, :, -chunk_len:]
        pred = pred.permute(1, 0, 2).contiguous().view(pred.shape[1], -1).unsqueeze(0)
        audio = self.vocoder(denorm_spec(pred))[0][0]
        pieces, p = [], 0
        while p < audio.shape[-1]:
            pieces.append(audio[p:p + chunk_len * up])
            p += chunk_len * up
        audio = self.sola_algorithm(pieces, ov * up)
        audio = audio[ov * up:-pad_len * up]      # as written in the reference (TTS.py:1600): empty when pad_len == 0
        out = []
        for n in lens:
            out.append(audio[:n * up])
            audio = audio[n * up:]
        return out

    def sola_algorithm(self, audio_fragments: List[torch.Tensor], overlap_len: int) -> torch.Tensor:
        """TTS.py:1611-1637 on the device (`gsv_sola`, csrc/sola.hip): correlation, argmax and cross-fade per
        neighbouring pair, then one compaction; only the stitched length returns to the host."""
        import ctypes as C
        from .. import _lib
        dev = self.configs.device
        dt = audio_fragments[0].dtype
        lens = [int(f.shape[0]) for f in audio_fragments]
        with torch.cuda.device(dev):
            buf = torch.cat([f.to(dev, torch.float32) for f in audio_fragments]).contiguous()
            out = torch.empty_like(buf)
            n_out = C.c_int(0)
            arr = (C.c_int * len(lens))(*lens)
            st = torch.cuda.current_stream(dev)
            _lib.check(_lib.lib().gsv_sola(buf.data_ptr(), arr, len(lens), int(overlap_len), out.data_ptr(), C.byref(n_out),
                                           C.c_void_p(st.cuda_stream)), "gsv_sola")
        return out[:n_out.value].to(dt)

    # ---- the pipeline (reference TTS.py:984-1365) ---------------------------------------------
    @torch.no_grad()
    def run(self, inputs: dict) -> Generator[Tuple[int, np.ndarray], None, None]:
        self.stop_flag = False
        top_k = inputs.get("top_k", 5)
        top_p = inputs.get("top_p", 1)
        temperature = inputs.get("temperature", 1)
        batch_size = inputs.get("batch_size", 1)
        batch_threshold = inputs.get("batch_threshold", 0.75)
        speed_factor = inputs.get("speed_factor", 1.0)
        split_bucket = inputs.get("split_bucket", True)
        return_fragment = inputs.get("return_fragment", False)
        fragment_interval = inputs.get("fragment_interval", 0.3)
        seed = inputs.get("seed", -1)
        seed = -1 if seed in ["", None] else seed
        actual_seed = set_seed(seed)
        parallel_infer = inputs.get("parallel_infer", True)
        repetition_penalty = inputs.get("repetition_penalty", 1.35)
        if fragment_interval < 0.01:
            fragment_interval = 0.01
        if return_fragment and split_bucket:
            split_bucket = False
        if speed_factor != 1.0:
            split_bucket = False
        elif getattr(self.configs, "use_vocoder", False) and parallel_infer:
            split_bucket = False           # v3 / v4 parallel runs are never bucketed (reference TTS.py:1060-1062)
        if inputs.get("super_sampling", False):
            # reference TTS.py:1040, 1408-1419: AP_BWE super-sampling, a separate model that is out of scope (DESIGN.md
            # section 7) -- refused, never silently ignored
            raise NotImplementedError("super_sampling (AP_BWE audio super-resolution) is not part of this engine")
        try:
            if self.t2s_model is None or self.vits_model is None:
                raise RuntimeError("init_t2s_weights / init_vits_weights first")
            # ---- reference audio and prompt text (reference TTS.py:1078-1120)
            ref_audio_path = inputs.get("ref_audio_path")
            prompt_text, prompt_lang = inputs.get("prompt_text"), inputs.get("prompt_lang", "")
            if ref_audio_path in [None, ""] and inputs.get("segments") is None and "text" in inputs and (
                    self.prompt_cache["prompt_semantic"] is None or self.prompt_cache["refer_spec"] in [None, []]):
                raise ValueError("ref_audio_path cannot be empty, when the reference audio is not set using set_ref_audio()")
            if ref_audio_path not in [None, ""] and ref_audio_path != self.prompt_cache["ref_audio_path"]:
                if not os.path.exists(ref_audio_path):
                    raise ValueError(f"{ref_audio_path} not exists")
                self.set_ref_audio(ref_audio_path)
            # auxiliary references for multi-speaker tone fusion (reference TTS.py:1098-1113): their spectrograms (and, v2Pro,
            # speaker embeddings) follow the main one; the style vector is the mean over all of them (models.py:971-985)
            aux = inputs.get("aux_ref_audio_paths") or []
            cached = self.prompt_cache.get("aux_ref_audio_paths") or []
            if "aux_ref_audio_paths" in inputs and not (len(set(aux) & set(cached)) == len(aux) == len(cached)):
                if self.prompt_cache["refer_spec"] in [None, []]:
                    raise ValueError("aux_ref_audio_paths need a main reference audio first (ref_audio_path / set_ref_audio)")
                self.prompt_cache["aux_ref_audio_paths"] = list(aux)
                self.prompt_cache["refer_spec"] = [self.prompt_cache["refer_spec"][0]]
                if self.prompt_cache.get("sv_emb"):
                    self.prompt_cache["sv_emb"] = [self.prompt_cache["sv_emb"][0]]
                for path in aux:
                    if path in [None, ""]:
                        continue
                    if not os.path.exists(path):
                        print("音频文件不存在，跳过：", path)
                        continue
                    spec_audio = self._get_ref_spec(path)
                    self.prompt_cache["refer_spec"].append(spec_audio)
                    if spec_audio[1] is not None:
                        self.prompt_cache["sv_emb"].append(self.sv_model.compute_embedding3(spec_audio[1]))
                self.vits_model.invalidate_refer()
            if prompt_text not in [None, ""]:
                from .text_segmentation_method import splits
                if prompt_lang not in self.configs.languages:
                    raise ValueError(f"prompt_lang {prompt_lang!r} is not one of {self.configs.languages}")
                prompt_text = prompt_text.strip("\n")
                if prompt_text[-1] not in splits:
                    prompt_text += "。" if prompt_lang != "en" else "."
                if self.prompt_cache["prompt_text"] != prompt_text:
                    phones, bert_features, norm_text = self.text_preprocessor.segment_and_extract_feature_for_text(
                        prompt_text, prompt_lang, self.configs.version)
                    self.prompt_cache.update(prompt_text=prompt_text, prompt_lang=prompt_lang, phones=phones,
                                             bert_features=bert_features, norm_text=norm_text)
            elif "prompt_text" in inputs and self.configs.use_vocoder:
                raise NO_PROMPT_ERROR("prompt_text cannot be empty when using SoVITS_V3")
            if not self.prompt_cache["refer_spec"] or (self.prompt_cache["prompt_semantic"] is None
                                                       and self.prompt_cache["phones"] is not None):
                raise NO_PROMPT_ERROR("set_prompt_cache() first (reference: ref_audio_path is required)")
            t0 = time.perf_counter()
            segments = inputs.get("segments")
            if segments is None:
                # reference TTS.py:1018-1024, 1100-1135: raw text through the TextPreprocessor (G2P back-ends are plug-ins,
                # gsv.text.cleaner.register_g2p); a `text_frontend` callable replaces it wholesale
                text, text_lang = inputs.get("text", ""), inputs.get("text_lang", "")
                method = inputs.get("text_split_method", "cut0")
                if self.text_frontend is not None:
                    segments = self.text_frontend(text, text_lang, method)
                else:
                    if text_lang not in self.configs.languages:
                        raise ValueError(f"text_lang {text_lang!r} is not one of {self.configs.languages}")
                    segments = self.text_preprocessor.preprocess(text, text_lang, method, self.configs.version)
            if len(segments) == 0:
                yield 16000, np.zeros(16000, dtype=np.int16)
                return
            prompt_data = None
            if self.prompt_cache["phones"] is not None:
                prompt_data = {"phones": self.prompt_cache["phones"], "bert_features": self.prompt_cache["bert_features"]}
            t1 = time.perf_counter()
            # token ids stay on the host here: the engines pack a whole batch and move it with ONE copy
            # (the reference's per-item .to(device), TTS.py:899-912, is ~75 tiny transfers per batch of 32)
            data, batch_index_list = self.to_batch(segments, prompt_data=prompt_data, batch_size=batch_size,
                                                   threshold=batch_threshold, split_bucket=split_bucket,
                                                   device=torch.device("cpu"), precision=self.precision)
            t2 = time.perf_counter()
            infer = (self.t2s_model.infer_panel_batch_infer if parallel_infer
                     else self.t2s_model.infer_panel_naive_batched)
            refer = [spec.to(device=self.configs.device) for spec, _ in self.prompt_cache["refer_spec"]]
            sv_kw = {"sv_emb": self.prompt_cache["sv_emb"]} if getattr(self.vits_model, "is_v2pro", False) else {}
            up = math.prod(self.vits_model.upsample_rates)
            audio, t_34, t_45 = [], 0.0, 0.0
            sr = self.configs.sampling_rate if not self.configs.use_vocoder else self.vocoder_configs["sr"]
            if self.configs.use_vocoder and self.vocoder is None:
                raise RuntimeError("init_vocoder() first")
            self.last_generated_tokens = 0
            for bi, item in enumerate(data):
                t3 = time.perf_counter()
                n = len(item["all_phones"])
                # no prompt text (reference TTS.py:1124-1131, 1223-1226): nothing is prepended and the AR decoder runs
                # prompt-free through the naive loop; v3/v4 require a prompt (TTS.py:1062-1063)
                no_prompt = prompt_data is None
                if no_prompt and self.configs.use_vocoder:
                    raise NO_PROMPT_ERROR("v3/v4 need the prompt text (phones) of the reference audio")
                prompt = None if no_prompt else self.prompt_cache["prompt_semantic"].view(1, -1).expand(n, -1)
                max_sec = self.configs.max_sec if self.configs.max_sec is not None else 54
                pred_list, idx_list = infer(item["all_phones"], item["all_phones_len"], prompt,
                                            item["all_bert_features"], top_k=top_k, top_p=top_p, temperature=temperature,
                                            early_stop_num=self.configs.hz * max_sec, max_len=item["max_len"],
                                            repetition_penalty=repetition_penalty, seed=actual_seed + bi)
                # stream-level wait only: the engine calls above already synchronised their own streams, and a DEVICE-wide
                # synchronize intermittently stalls 20-30 ms on this ROCm build
                torch.cuda.current_stream(self.configs.device).synchronize()
                t4 = time.perf_counter()
                t_34 += t4 - t3
                if no_prompt:     # idx is reported as 0 and y holds only generated tokens (t2s_model.py:916-917)
                    pred = list(pred_list)
                    idx_list = [int(p.shape[0]) for p in pred]
                else:
                    pred = [p[-i:] if i > 0 else p[:0] for p, i in zip(pred_list, idx_list)]
                self.last_generated_tokens += int(sum(idx_list))
                frags: List[torch.Tensor] = []
                if self.configs.use_vocoder:
                    # TTS.py:1283-1299
                    sample_steps = inputs.get("sample_steps", 32)
                    dev_ph = [ph.to(self.configs.device) for ph in item["phones"]]
                    if parallel_infer:
                        frags = self.using_vocoder_synthesis_batched_infer(idx_list, pred_list, dev_ph, speed=speed_factor,
                                                                           sample_steps=sample_steps, seed=actual_seed + bi)
                    else:
                        for k, idx in enumerate(idx_list):
                            frags.append(self.using_vocoder_synthesis(pred_list[k][-idx:].view(1, 1, -1), dev_ph[k].view(1, -1),
                                                                      speed=speed_factor, sample_steps=sample_steps,
                                                                      seed=actual_seed + bi * 4096 + k))
                elif speed_factor == 1.0:
                    # one decode over the batch folded into the time axis (TTS.py:1259-1282)
                    ends = np.cumsum([0] + [int(p.shape[0]) * 2 * up for p in pred])
                    keep = [k for k, p in enumerate(pred) if p.shape[0] > 0]
                    if keep:
                        all_pred = torch.cat([pred[k] for k in keep]).view(1, 1, -1)
                        all_ph = torch.cat([item["phones"][k] for k in keep]).view(1, -1)
                        wav = self.vits_model.decode(all_pred, all_ph, refer, speed=speed_factor,
                                                     seed=actual_seed + bi, **sv_kw)[0, 0]
                    else:
                        wav = torch.zeros(0, dtype=self.precision, device=self.configs.device)
                    o = 0
                    for k, p in enumerate(pred):
                        nsm = int(p.shape[0]) * 2 * up
                        frags.append(wav[o:o + nsm])
                        o += nsm
                else:
                    for k, p in enumerate(pred):
                        frags.append(self.vits_model.decode(p.view(1, 1, -1), item["phones"][k].view(1, -1), refer,
                                                            speed=speed_factor, seed=actual_seed + bi, **sv_kw)[0, 0])
                # stream-level wait only: the engine calls above already synchronised their own streams, and a DEVICE-wide
                # synchronize intermittently stalls 20-30 ms on this ROCm build
                torch.cuda.current_stream(self.configs.device).synchronize()
                t5 = time.perf_counter()
                t_45 += t5 - t4
                if return_fragment:
                    yield self.audio_postprocess([frags], sr, None, speed_factor, False, fragment_interval)
                else:
                    audio.append(frags)
                if self.stop_flag:
                    yield 16000, np.zeros(16000, dtype=np.int16)
                    return
            self.last_timing = (t1 - t0, t2 - t1, t_34, t_45)
            if not return_fragment:
                if len(audio) == 0:
                    yield 16000, np.zeros(16000, dtype=np.int16)
                    return
                t6 = time.perf_counter()
                result = self.audio_postprocess(audio, sr, batch_index_list, speed_factor, split_bucket, fragment_interval)
                self.last_postprocess_s = time.perf_counter() - t6
                yield result
        except Exception as e:
            traceback.print_exc()
            # the reference yields 1 s of silence, rebuilds both models, then re-raises (TTS.py:1352-1363)
            yield 16000, np.zeros(16000, dtype=np.int16)
            try:
                if self._t2s_state is not None and self._vits_state is not None:
                    self.t2s_model = None
                    self.vits_model = None
                    self.init_t2s_weights(self.configs.t2s_weights_path, state=self._t2s_state)
                    self.init_vits_weights(self.configs.vits_weights_path, state=self._vits_state)
            finally:
                raise e

