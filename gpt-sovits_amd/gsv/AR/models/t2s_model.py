"""Host-side mirror of the reference's `Text2SemanticDecoder` inference interface
(reference GPT_SoVITS/AR/models/t2s_model.py:260-935), backed by the HIP engine.

Same entry points, argument meaning and return values as the reference:
`infer_panel_batch_infer` (:583), `infer_panel_naive_batched` (:781), `infer_panel_naive` (:814),
`infer_panel` (:920).  All arithmetic (embeddings, prefill, KV-cached decode, sampling, EOS /
early-stop bookkeeping) runs in libgsv_hip.so; this class only packs inputs and unpacks
outputs.  There is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ... import _lib


def _sine_pe(n_pos: int, dim: int) -> torch.Tensor:
    """Position table, computed exactly as the reference does on the host
    (AR/modules/embedding.py:54-72) so the fp32 engine sees bit-identical values."""
    position = torch.arange(0, n_pos, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(n_pos, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _host_nonzero(t: torch.Tensor) -> bool:
    """any(t != 0) for a host tensor without a torch dispatch (bfloat16 has no numpy view: torch then)"""
    try:
        return bool(np.any(t.detach().numpy()))
    except (TypeError, RuntimeError):
        return bool(t.any())


class Text2SemanticDecoder:
    def __init__(self, config: dict, device="cuda:0", dtype=torch.float16, max_batch: int = 32,
                 max_seq: int = 2048, norm_first: bool = False, top_k: int = 3):
        m = config["model"]
        self.config = config
        self.model_dim = m["hidden_dim"]
        self.embedding_dim = m["embedding_dim"]
        self.num_head = m["head"]
        self.num_layers = m["n_layer"]
        self.vocab_size = m["vocab_size"]
        self.phoneme_vocab_size = m["phoneme_vocab_size"]
        self.EOS = m["EOS"]
        assert self.EOS == self.vocab_size - 1
        assert not norm_first, "the reference's inference checkpoints are post-LN"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gsv Text2SemanticDecoder runs on an MI355X (cuda/HIP device) only; "
                               "there is no CPU path in the product")
        self.dtype = dtype
        self.max_batch = max_batch
        self.max_seq = max_seq
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        with torch.cuda.device(self.device):
            _lib.init(idx)
            cfg = _lib.T2SConfig(self.num_layers, self.model_dim, self.num_head, 4 * self.model_dim,
                                 self.vocab_size, self.phoneme_vocab_size, 1024)
            h = C.c_void_p()
            _lib.check(_lib.lib().gsv_t2s_create(C.byref(cfg), _lib.dtype_code(dtype), max_batch, max_seq,
                                                 C.byref(h)), "gsv_t2s_create")
            self._h = h
            self.stream = torch.cuda.Stream(device=self.device)
        self._loaded = False
        self.infer_panel = self.infer_panel_naive

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().gsv_t2s_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- weights -------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts the reference checkpoint's `weight` dict (keys with or without the
        Lightning `model.` prefix, reference TTS_infer_pack/TTS.py:590-603)."""
        if self._loaded:
            raise RuntimeError("weights already loaded; create a new Text2SemanticDecoder")
        l = _lib.lib()
        with torch.cuda.device(self.device):
            for k, v in state_dict.items():
                if not torch.is_tensor(v):
                    continue
                t = v.detach().to("cpu", torch.float32).contiguous()
                _lib.check(l.gsv_t2s_load_tensor(self._h, k.encode(), t.data_ptr(), t.numel()), f"load {k}")
            pe = _sine_pe(4000, self.embedding_dim).contiguous()
            _lib.check(l.gsv_t2s_load_tensor(self._h, b"pe", pe.data_ptr(), pe.numel()), "load pe")
            _lib.check(l.gsv_t2s_finalize(self._h), "gsv_t2s_finalize")
        self._loaded = True
        return self

    # ---- engine call ---------------------------------------------------------------
    def _run(self, x: Sequence[torch.Tensor], prompts: torch.Tensor, bert: Sequence[torch.Tensor], top_k, top_p,
             early_stop_num, temperature, repetition_penalty, eos_mask_steps, noise=None, seed=0,
             max_steps: int = 1500, force_tokens=None, dump_logits=False):
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        B = len(x)
        dev = self.device
        if prompts is None:                  # prompt-free (reference t2s_model.py:849-856): empty audio prefix
            prompts = torch.zeros(B, 0, dtype=torch.int64)
        l = _lib.lib()
        with torch.cuda.device(dev):
            lens = [int(t.shape[-1]) for t in x]
            P = int(prompts.shape[1])
            need = max(lens) + P + 2
            wanted = max_steps if early_stop_num in (-1, None) else min(max_steps, int(early_stop_num) + 1)
            budget = min(wanted, self.max_seq - need)
            if budget < 1:
                raise ValueError(f"sequence of {need} positions does not fit max_seq={self.max_seq}")
            arena_bound = budget < wanted
            phones = torch.cat([t.reshape(-1) for t in x]).to(dev, torch.int32).contiguous()
            lens_h = (C.c_int32 * B)(*lens)
            bert_dev = None
            if bert is not None and any(b_ is not None for b_ in bert):
                # None or all-zero features (non-zh text): bert_proj(0) is its bias, handled in the engine.  Host tensors are
                # tested on the host and travel as ONE copy (the pipeline hands over 32 zero blocks of 330 KB per batch: as 32
                # pageable uploads + a device-side any() + sync they were ~1 ms of idle GPU in front of every prefill)
                on_host = all(b_ is None or b_.device.type == "cpu" for b_ in bert)
                if on_host:
                    # numpy on the host view: a torch reduction costs ~1.3 ms per block on a many-core host (thread wake-up), i.e.
                    # 40 ms per batch of 32 (tools/prefill_probe.py)
                    if any(b_ is not None and _host_nonzero(b_) for b_ in bert):
                        cols = [(torch.zeros(lens[i], 1024) if b_ is None else b_.reshape(1024, -1).t().float())
                                for i, b_ in enumerate(bert)]
                        bert_dev = torch.cat(cols, 0).contiguous().to(dev)
                else:
                    cols = [(torch.zeros(lens[i], 1024, device=dev) if b_ is None
                             else b_.reshape(1024, -1).t().to(dev, torch.float32)) for i, b_ in enumerate(bert)]
                    allb = torch.cat(cols, 0).contiguous()
                    if bool(torch.any(allb)):            # one host sync for the whole batch
                        bert_dev = allb
            pr = prompts.to(dev, torch.int32).contiguous()
            out_tokens = torch.zeros(B, budget, dtype=torch.int32, device=dev)
            out_len = torch.full((B,), -1, dtype=torch.int32, device=dev)
            noise_dev, noise_rows = None, 0
            if noise is not None:
                noise_dev = noise.to(dev, torch.float32).contiguous()
                assert noise_dev.dim() == 3 and noise_dev.shape[2] == self.vocab_size and noise_dev.shape[0] >= budget
                noise_rows = noise_dev.shape[1]
                noise_dev = noise_dev[:budget].contiguous()
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            sp = _lib.SamplingParams(int(top_k) if top_k is not None else 0, float(top_p if top_p is not None else 1.0),
                                     float(temperature), float(repetition_penalty),
                                     -1 if early_stop_num is None else int(early_stop_num), int(eos_mask_steps),
                                     int(budget), int(seed) & 0xFFFFFFFFFFFFFFFF)
            if early_stop_num not in (-1, None) and budget < int(early_stop_num) + 1:
                sp.early_stop_num = -1   # the arena bound (max_steps) ends generation first
            s = C.c_void_p(self.stream.cuda_stream)
            _lib.check(l.gsv_t2s_prefill(self._h, phones.data_ptr(), C.cast(lens_h, C.c_void_p), B,
                                         bert_dev.data_ptr() if bert_dev is not None else None,
                                         pr.data_ptr() if P > 0 else None, P, s),
                       "gsv_t2s_prefill")
            steps = C.c_int(0)
            dump = None
            if force_tokens is not None or dump_logits:
                # parity hooks (tests): teacher forcing and the per-step logits of whichever decode path runs
                ft = None
                if force_tokens is not None:
                    ft = torch.zeros(B, budget, dtype=torch.int32, device=dev)
                    src = force_tokens.to(dev, torch.int32)
                    n = min(budget, int(src.shape[1]))
                    ft[:, :n] = src[:, :n]
                drawn = None
                if dump_logits:
                    dump = torch.zeros(budget, B, self.vocab_size, dtype=torch.float32, device=dev)
                    drawn = torch.full((budget, B, 2), -1, dtype=torch.int32, device=dev)
                torch.cuda.current_stream(dev).synchronize()
                _lib.check(l.gsv_t2s_set_debug(self._h, ft.data_ptr() if ft is not None else None,
                                               dump.data_ptr() if dump is not None else None,
                                               drawn.data_ptr() if drawn is not None else None), "gsv_t2s_set_debug")
                self.last_drawn_dump = drawn
            self.last_logits_dump = dump
            _lib.check(l.gsv_t2s_decode(self._h, C.byref(sp), noise_dev.data_ptr() if noise_dev is not None else None,
                                        noise_rows, out_tokens.data_ptr(), out_len.data_ptr(), C.byref(steps), s),
                       "gsv_t2s_decode")
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            idx = out_len.cpu().tolist()
            self.last_steps = steps.value
            y_all = torch.cat([pr, out_tokens], dim=1).long()       # one concat for the batch; rows are sliced as views
            y_list = []
            self.last_truncated = [b for b in range(B) if idx[b] < 0] if arena_bound else []
            if self.last_truncated:
                # the reference always allows 1500 steps (t2s_model.py:694) or early_stop_num; here the K/V arena ended
                # these rows first -- never silently
                warnings.warn(f"Text2SemanticDecoder: rows {self.last_truncated} were cut after {steps.value - 1} tokens by the "
                              f"K/V arena (max_seq={self.max_seq}, {need} positions used by text + prompt); create the "
                              f"decoder with a larger max_seq", RuntimeWarning, stacklevel=3)
            for b in range(B):
                n = idx[b] if idx[b] >= 0 else steps.value - 1
                idx[b] = n
                y_list.append(y_all[b, :P + n])
        return y_list, idx

    # ---- reference entry points -----------------------------------------------------
    @torch.no_grad()
    def infer_panel_batch_infer(self, x: List[torch.LongTensor], x_lens: torch.LongTensor, prompts: torch.LongTensor,
                                bert_feature: List[torch.Tensor], top_k: int = -100, top_p: int = 100,
                                early_stop_num: int = -1, temperature: float = 1.0,
                                repetition_penalty: float = 1.35, **kwargs):
        """reference t2s_model.py:583-779: returns (y_list, idx_list); y_list[i] = prompt + generated
        tokens (finishing token dropped), idx_list[i] = number of generated tokens."""
        if prompts is None:
            return self.infer_panel_naive_batched(x, x_lens, prompts, bert_feature, top_k=top_k, top_p=top_p,
                                                  early_stop_num=early_stop_num, temperature=temperature, **kwargs)
        out = []
        ys: List[Optional[torch.Tensor]] = [None] * len(x)
        idxs: List[Optional[int]] = [None] * len(x)
        noise = kwargs.get("noise")
        for lo in range(0, len(x), self.max_batch):
            hi = min(lo + self.max_batch, len(x))
            nz = noise
            if noise is not None and noise.shape[1] > 1:
                nz = noise[:, lo:hi]
            y, i = self._run(x[lo:hi], prompts[lo:hi], bert_feature[lo:hi], top_k, top_p, early_stop_num, temperature,
                             repetition_penalty, eos_mask_steps=1, noise=nz, seed=kwargs.get("seed", 0),
                             max_steps=kwargs.get("max_steps", 1500),
                             force_tokens=None if kwargs.get("force_tokens") is None else kwargs["force_tokens"][lo:hi],
                             dump_logits=kwargs.get("dump_logits", False))
            ys[lo:hi] = y
            idxs[lo:hi] = i
        return ys, idxs

    @torch.no_grad()
    def infer_panel_naive(self, x: torch.LongTensor, x_lens: torch.LongTensor, prompts: torch.LongTensor,
                          bert_feature: torch.Tensor, top_k: int = -100, top_p: int = 100, early_stop_num: int = -1,
                          temperature: float = 1.0, repetition_penalty: float = 1.35, **kwargs):
        """reference t2s_model.py:814-918: batch 1, EOS masked while idx < 11; returns (y[:, :-1], idx)."""
        y, i = self._run([x[0]], prompts, [bert_feature[0] if bert_feature is not None else None], top_k, top_p, early_stop_num, temperature,
                         repetition_penalty, eos_mask_steps=11, noise=kwargs.get("noise"), seed=kwargs.get("seed", 0),
                         max_steps=kwargs.get("max_steps", 1500))
        return y[0].unsqueeze(0), (0 if prompts is None else i[0])      # prompt-free reports idx 0 (t2s_model.py:916-917)

    @torch.no_grad()
    def infer_panel_naive_batched(self, x, x_lens, prompts, bert_feature, top_k: int = -100, top_p: int = 100,
                                  early_stop_num: int = -1, temperature: float = 1.0,
                                  repetition_penalty: float = 1.35, **kwargs):
        """reference t2s_model.py:781-812: the naive loop per item."""
        y_list, idx_list = [], []
        for i in range(len(x)):
            y, idx = self.infer_panel_naive(x[i].unsqueeze(0), x_lens[i] if x_lens is not None else None,
                                            prompts[i].unsqueeze(0) if prompts is not None else None,
                                            [bert_feature[i]] if bert_feature[i] is None else bert_feature[i].unsqueeze(0),
                                            top_k, top_p, early_stop_num, temperature,
                                            repetition_penalty, **kwargs)
            y_list.append(y[0])
            idx_list.append(idx)
        return y_list, idx_list

    # ---- decode-engine selection / report -------------------------------------------
    def set_mega(self, on: bool):
        """A/B switch: False makes later calls use the launch-per-phase decode step instead of the persistent engine
        (csrc/t2s_mega.hip; fp16, v1/v2 shape, batch <= 128 in one launch)."""
        _lib.check(_lib.lib().gsv_t2s_set_mega(self._h, int(bool(on))), "gsv_t2s_set_mega")

    def debug_stall(self, member: int):
        """test hook: the next persistent launch loses one hand-off publish of `member` (group 0)"""
        _lib.check(_lib.lib().gsv_t2s_debug_stall(self._h, int(member)), "gsv_t2s_debug_stall")

    def engine_stats(self):
        """(engine_available, fallbacks, (epoch, workgroup, hop code) of the last hand-off timeout): a persistent launch
        that times out is re-run on the launch-per-phase step and the handle stops using the engine."""
        av, fb, e3 = C.c_int(0), C.c_int(0), (C.c_uint * 3)()
        _lib.check(_lib.lib().gsv_t2s_engine_stats(self._h, C.byref(av), C.byref(fb), e3), "gsv_t2s_engine_stats")
        return bool(av.value), fb.value, tuple(e3)

    def decode_info(self):
        """(mode, device_ms, steps) of the last decode call: mode 1 = persistent engine, 0 = launch per phase."""
        mode, ms, steps = C.c_int(0), C.c_float(0), C.c_int(0)
        _lib.check(_lib.lib().gsv_t2s_decode_info(self._h, C.byref(mode), C.byref(ms), C.byref(steps)), "gsv_t2s_decode_info")
        return mode.value, ms.value, steps.value

    # ---- measurement hooks (bench.py) ----------------------------------------------
    def debug_logits(self, B: int) -> torch.Tensor:
        out = torch.empty(B, self.vocab_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gsv_t2s_debug_logits(self._h, out.data_ptr(), C.c_void_p(self.stream.cuda_stream)))
            self.stream.synchronize()
        return out

    def time_attention(self, iters: int = 20):
        """(avg ms per decode-attention launch, algorithmic HBM bytes per launch, algorithmic bytes of a
        whole decode step, eager ms of one step's 24-layer kernel sequence) at the current cache state,
        measured in situ with HIP events on the engine stream (gsv_t2s_time_step)."""
        ms = C.c_float(0)
        step = C.c_float(0)
        ab = C.c_int64(0)
        with torch.cuda.device(self.device):
            l = _lib.lib()
            _lib.check(l.gsv_t2s_time_step(self._h, iters, C.byref(step), C.byref(ms), C.c_void_p(self.stream.cuda_stream)))
            total = l.gsv_t2s_step_bytes(self._h, C.byref(ab))
        return ms.value, ab.value, total, step.value
