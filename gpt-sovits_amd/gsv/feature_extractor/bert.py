"""zh BERT features on the HIP library (reference TTS_infer_pack/TextPreprocessor.py:191-204, TTS.py:472-482:
`AutoModelForMaskedLM` = chinese-roberta-wwm-ext-large, a BERT-large encoder; the pipeline takes `hidden_states[-3]` -- the
output of encoder layer 22 of 24 -- and drops [CLS] / [SEP], one 1024-vector per character of the normalised text).

`BertFeature(text) -> Tensor[len(text), 1024]` is the `bert_fn` plug-in of gsv.TTS_infer_pack.TextPreprocessor.
Tokenisation: one token per character through vocab.txt (lower-cased, unknown -> [UNK]); that is what BERT's tokenizer yields
for CJK text and punctuation, and the only case the reference accepts (it asserts len(word2ph) == len(text)).
Embedding rows are gathered and summed on the host (<= 512 x 1024 values); the 22 encoder layers run as `gsv_op_conv1d` /
`gsv_op_flash_attn64` / `gsv_op_layernorm` calls in fp16 with fp32 accumulation and normalisation statistics.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from .. import _lib

ACT_NONE, ACT_GELU = 0, 7


class BertFeature:
    def __init__(self, base_path: Optional[str] = None, device="cuda:0", state_dict: Optional[dict] = None,
                 vocab: Optional[List[str]] = None, layers_used: int = 22, heads: int = 16, eps: float = 1e-12):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gsv BertFeature runs on an MI355X (cuda/HIP device) only; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.init(idx)
        if state_dict is None:
            if base_path is None or not os.path.exists(base_path):
                raise FileNotFoundError(base_path)
            st = os.path.join(base_path, "model.safetensors")
            if os.path.exists(st):
                from safetensors.torch import load_file
                state_dict = load_file(st)
            else:
                state_dict = torch.load(os.path.join(base_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
            with open(os.path.join(base_path, "vocab.txt"), encoding="utf-8") as f:
                vocab = [ln.rstrip("\n") for ln in f]
        if vocab is None:
            raise ValueError("vocab (list of tokens, index = id) is required with an in-memory state_dict")
        self.tok: Dict[str, int] = {t: i for i, t in enumerate(vocab)}
        for need in ("[CLS]", "[SEP]", "[UNK]"):
            if need not in self.tok:
                raise ValueError(f"vocab lacks {need}")
        self.layers_used, self.heads, self.eps = layers_used, heads, eps
        sd = {}
        for k, v in state_dict.items():
            if not torch.is_tensor(v):
                continue
            for pre in ("bert.", "roberta."):
                if k.startswith(pre):
                    k = k[len(pre):]
            sd[k] = v.detach().float().cpu()
        self.emb_word = sd["embeddings.word_embeddings.weight"]
        self.emb_pos = sd["embeddings.position_embeddings.weight"]
        self.emb_type0 = sd["embeddings.token_type_embeddings.weight"][0]
        self.hidden = int(self.emb_word.shape[1])
        if self.hidden // heads != 64:
            raise NotImplementedError("head dim must be 64")
        dev = self.device
        self.w: Dict[str, torch.Tensor] = {}

        def put(name, t, half=True):
            self.w[name] = t.to(dev, torch.float16 if half else torch.float32).contiguous()
        put("emb_ln_w", sd["embeddings.LayerNorm.weight"], False)
        put("emb_ln_b", sd["embeddings.LayerNorm.bias"], False)
        for i in range(layers_used):
            p = f"encoder.layer.{i}."
            put(f"l{i}.qkv_w", torch.cat([sd[p + "attention.self.query.weight"], sd[p + "attention.self.key.weight"],
                                           sd[p + "attention.self.value.weight"]], 0))
            put(f"l{i}.qkv_b", torch.cat([sd[p + "attention.self.query.bias"], sd[p + "attention.self.key.bias"],
                                           sd[p + "attention.self.value.bias"]], 0), False)
            put(f"l{i}.o_w", sd[p + "attention.output.dense.weight"])
            put(f"l{i}.o_b", sd[p + "attention.output.dense.bias"], False)
            put(f"l{i}.ln1_w", sd[p + "attention.output.LayerNorm.weight"], False)
            put(f"l{i}.ln1_b", sd[p + "attention.output.LayerNorm.bias"], False)
            put(f"l{i}.f1_w", sd[p + "intermediate.dense.weight"])
            put(f"l{i}.f1_b", sd[p + "intermediate.dense.bias"], False)
            put(f"l{i}.f2_w", sd[p + "output.dense.weight"])
            put(f"l{i}.f2_b", sd[p + "output.dense.bias"], False)
            put(f"l{i}.ln2_w", sd[p + "output.LayerNorm.weight"], False)
            put(f"l{i}.ln2_b", sd[p + "output.LayerNorm.bias"], False)
        self.ffn = int(self.w["l0.f1_w"].shape[0])

    def tokenize(self, text: str) -> List[int]:
        unk = self.tok["[UNK]"]
        return [self.tok["[CLS]"]] + [self.tok.get(ch.lower(), unk) for ch in text] + [self.tok["[SEP]"]]

    def _gemm(self, st, x, T, Cin, w, Cout, bias, act=ACT_NONE, res=None):
        y = torch.empty(T, Cout, dtype=torch.float16, device=self.device)
        d = _lib.ConvDesc()
        d.x, d.w, d.y, d.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
        d.res = res.data_ptr() if res is not None else None
        d.T_in = d.T_out = T
        d.Cin, d.Cout, d.taps, d.stride, d.dil, d.pad = Cin, Cout, 1, 1, 1, 0
        d.post_act, d.scale = act, 1.0
        _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.GSV_F16, st), "gsv_op_conv1d")
        return y

    def _ln(self, st, x, T, w, b):
        y = torch.empty(T, self.hidden, dtype=torch.float16, device=self.device)
        _lib.check(_lib.lib().gsv_op_layernorm(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), y.data_ptr(), T, self.hidden,
                                               self.eps, _lib.GSV_F16, st), "gsv_op_layernorm")
        return y

    @torch.no_grad()
    def __call__(self, text: str) -> torch.Tensor:
        ids = self.tokenize(text)
        T = len(ids)
        if T > self.emb_pos.shape[0]:
            raise ValueError(f"{T} tokens exceed BERT's {self.emb_pos.shape[0]} positions (the front-end splits above 510 characters)")
        h, w, l, dev = self.hidden, self.w, _lib.lib(), self.device
        emb = self.emb_word[torch.tensor(ids)] + self.emb_pos[:T] + self.emb_type0
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            x = self._ln(st, emb.to(dev, torch.float16).contiguous(), T, w["emb_ln_w"], w["emb_ln_b"])
            vt = torch.empty(h * ((T + 31) // 32 * 32), dtype=torch.float16, device=dev)
            for i in range(self.layers_used):
                qkv = self._gemm(st, x, T, h, w[f"l{i}.qkv_w"], 3 * h, w[f"l{i}.qkv_b"])
                att = torch.empty(T, h, dtype=torch.float16, device=dev)
                _lib.check(l.gsv_op_flash_attn64(qkv.data_ptr(), T, self.heads, 0.125, vt.data_ptr(), att.data_ptr(), st),
                           "gsv_op_flash_attn64")
                o = self._gemm(st, att, T, h, w[f"l{i}.o_w"], h, w[f"l{i}.o_b"], res=x)
                x = self._ln(st, o, T, w[f"l{i}.ln1_w"], w[f"l{i}.ln1_b"])
                f = self._gemm(st, x, T, h, w[f"l{i}.f1_w"], self.ffn, w[f"l{i}.f1_b"], act=ACT_GELU)
                f2 = self._gemm(st, f, T, self.ffn, w[f"l{i}.f2_w"], h, w[f"l{i}.f2_b"], res=x)
                x = self._ln(st, f2, T, w[f"l{i}.ln2_w"], w[f"l{i}.ln2_b"])
        return x[1:-1].float()
