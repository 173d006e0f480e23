"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's AR
semantic-token decoder.  Never imported by the product path (gsv/); only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Pinned against the reference itself: oracle/gen_golden.py imports
/root/reference's Text2SemanticDecoder in the build container, loads the same
synthetic checkpoint, and requires identical token ids (tests/golden/t2s_*.npz).

Plain torch fp32 CPU tensor ops, written from the algorithm, no reference code:
  * embeddings            reference AR/modules/embedding.py:8-78, t2s_model.py:612-622,636-641
  * prefill (process_prompt)   reference AR/models/t2s_model.py:135-174, 230-243
  * decode step                reference AR/models/t2s_model.py:176-221, 245-257
  * batched loop (H1)          reference AR/models/t2s_model.py:583-779
  * naive loop (H1b)           reference AR/models/t2s_model.py:814-918
  * sampling (H5)              reference AR/models/utils.py:140-199
Differences in *form*: ragged rows are kept unpadded (left-pad keys are masked
out in the reference, t2s_model.py:644-683, so they contribute exactly zero), the KV
cache is a preallocated buffer (the reference re-concatenates, :186-187), and
finished rows are flagged instead of index_select-ed away (:727-745).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


def sine_pe_table(n_pos: int, dim: int) -> torch.Tensor:
    """reference AR/modules/embedding.py:54-72 (non-reversed branch)."""
    pos = torch.arange(0, n_pos, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(n_pos, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# ----------------------------------------------------------------------------
# sampling (reference AR/models/utils.py:147-199)
# ----------------------------------------------------------------------------
def apply_repetition_penalty(logits: torch.Tensor, previous_tokens: Optional[torch.Tensor],
                             repetition_penalty: float) -> torch.Tensor:
    """reference utils.py:159-167.  NOTE the reference does this IN PLACE on the caller's
    logits (`logits.scatter_`), so the later `argmax(logits)` EOS test of
    t2s_model.py:721 sees the *penalised* logits; callers here use the return value."""
    logits = logits.clone()
    if previous_tokens is not None and repetition_penalty != 1.0:
        prev = previous_tokens.long()
        score = torch.gather(logits, 1, prev)
        score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
        logits.scatter_(1, prev, score)
    return logits


def logits_to_probs(logits: torch.Tensor, previous_tokens: Optional[torch.Tensor], temperature: float = 1.0,
                    top_k: Optional[int] = None, top_p: Optional[float] = None,
                    repetition_penalty: float = 1.0) -> torch.Tensor:
    logits = apply_repetition_penalty(logits, previous_tokens, repetition_penalty)
    if top_p is not None and top_p < 1.0:
        sl, si = torch.sort(logits, descending=True)
        cum = torch.cumsum(torch.softmax(sl, dim=-1), dim=-1)
        rm = cum > top_p
        rm[:, 0] = False
        rm = rm.scatter(1, si, rm)
        logits = logits.masked_fill(rm, -float("inf"))
    logits = logits / max(temperature, 1e-5)
    if top_k is not None:
        v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
        pivot = v[:, -1].unsqueeze(-1)
        logits = torch.where(logits < pivot, torch.full_like(logits, -float("inf")), logits)
    return torch.softmax(logits, dim=-1)


def sample(logits: torch.Tensor, previous_tokens: Optional[torch.Tensor], noise: Optional[torch.Tensor] = None,
           **kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """`noise` = the Exp(1) draws `q` of reference utils.py:143 (injectable so the
    HIP path and the oracle can be fed identical randomness); None -> torch RNG."""
    probs = logits_to_probs(logits, previous_tokens, **kw)
    q = torch.empty_like(probs).exponential_(1) if noise is None else noise[:, : probs.shape[1]]
    idx = torch.argmax(probs / q, dim=-1, keepdim=True).to(torch.int)
    return idx, probs


# ----------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------
class T2SOracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: dict, dtype=torch.float32):
        m = config["model"]
        self.d = m["hidden_dim"]
        self.H = m["head"]
        self.L = m["n_layer"]
        self.V = m["vocab_size"]
        self.EOS = m["EOS"]
        sd = {}
        for k, v in state_dict.items():
            k = k[6:] if k.startswith("model.") else k
            sd[k] = v.detach().to(dtype).cpu()
        self.sd = sd
        self.dtype = dtype
        self.pe = sine_pe_table(4000, self.d).to(dtype)
        self.alpha_t = sd["ar_text_position.alpha"].item()
        self.alpha_a = sd["ar_audio_position.alpha"].item()
        self.layers = []
        for i in range(self.L):
            p = f"h.layers.{i}."
            self.layers.append(dict(
                qkv_w=sd[p + "self_attn.in_proj_weight"], qkv_b=sd[p + "self_attn.in_proj_bias"],
                out_w=sd[p + "self_attn.out_proj.weight"], out_b=sd[p + "self_attn.out_proj.bias"],
                w1=sd[p + "linear1.weight"], b1=sd[p + "linear1.bias"],
                w2=sd[p + "linear2.weight"], b2=sd[p + "linear2.bias"],
                n1w=sd[p + "norm1.weight"], n1b=sd[p + "norm1.bias"],
                n2w=sd[p + "norm2.weight"], n2b=sd[p + "norm2.bias"]))

    # -- embeddings (H2) ------------------------------------------------------
    def embed_text(self, ids: torch.Tensor, bert: torch.Tensor) -> torch.Tensor:
        """ids [X] int64, bert [1024, X] -> [X, d]   (t2s_model.py:612-617)"""
        sd = self.sd
        x = sd["ar_text_embedding.word_embeddings.weight"][ids]
        x = x + F.linear(bert.to(self.dtype).t(), sd["bert_proj.weight"], sd["bert_proj.bias"])
        return x + self.alpha_t * self.pe[: x.shape[0]]

    def embed_audio(self, y: torch.Tensor, start: int = 0) -> torch.Tensor:
        """y [P] -> [P, d] with positions start.. (t2s_model.py:636-640, 766-769)"""
        e = self.sd["ar_audio_embedding.word_embeddings.weight"][y]
        return e + self.alpha_a * self.pe[start: start + y.shape[0]]

    # -- one transformer layer on a single sequence --------------------------
    def _post(self, lay, x, attn):
        x = x + F.linear(attn, lay["out_w"], lay["out_b"])
        x = F.layer_norm(x, [self.d], lay["n1w"], lay["n1b"], 1e-5)
        h = F.linear(F.relu(F.linear(x, lay["w1"], lay["b1"])), lay["w2"], lay["b2"])
        return F.layer_norm(x + h, [self.d], lay["n2w"], lay["n2b"], 1e-5)

    def prefill_one(self, xy: torch.Tensor, x_len: int):
        """xy [S, d] (x rows then prompt rows) -> h [S, d], K/V lists [L][S, d].
        Mask (t2s_model.py:655-683): x rows see x keys only; y rows see all x and causal y."""
        S = xy.shape[0]
        allow = torch.zeros(S, S, dtype=torch.bool)
        allow[:x_len, :x_len] = True
        allow[x_len:, :x_len] = True
        allow[x_len:, x_len:] = torch.tril(torch.ones(S - x_len, S - x_len, dtype=torch.bool))
        bias = torch.zeros(S, S, dtype=self.dtype).masked_fill(~allow, -float("inf"))
        ks, vs = [], []
        x = xy
        hd = self.d // self.H
        for lay in self.layers:
            q, k, v = F.linear(x, lay["qkv_w"], lay["qkv_b"]).chunk(3, dim=-1)
            ks.append(k)
            vs.append(v)
            qh = q.view(S, self.H, hd).transpose(0, 1)
            kh = k.view(S, self.H, hd).transpose(0, 1)
            vh = v.view(S, self.H, hd).transpose(0, 1)
            w = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd) + bias, dim=-1)
            attn = (w @ vh).transpose(0, 1).reshape(S, self.d)
            x = self._post(lay, x, attn)
        return x, ks, vs

    def decode_one(self, x: torch.Tensor, kc: List[torch.Tensor], vc: List[torch.Tensor], n: int):
        """x [1, d]; kc/vc [L][Smax, d] with n valid rows; appends row n.  (t2s_model.py:176-221)"""
        hd = self.d // self.H
        for li, lay in enumerate(self.layers):
            q, k, v = F.linear(x, lay["qkv_w"], lay["qkv_b"]).chunk(3, dim=-1)
            kc[li][n] = k[0]
            vc[li][n] = v[0]
            kh = kc[li][: n + 1].view(n + 1, self.H, hd).transpose(0, 1)
            vh = vc[li][: n + 1].view(n + 1, self.H, hd).transpose(0, 1)
            qh = q.view(1, self.H, hd).transpose(0, 1)
            w = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1)
            attn = (w @ vh).transpose(0, 1).reshape(1, self.d)
            x = self._post(lay, x, attn)
        return x

    def logits(self, h: torch.Tensor) -> torch.Tensor:
        return F.linear(h, self.sd["ar_predict_layer.weight"])

    # -- H1: batched loop ------------------------------------------------------
    @torch.no_grad()
    def infer_panel_batch_infer(self, x: Sequence[torch.Tensor], x_lens, prompts: torch.Tensor,
                                bert_feature: Sequence[torch.Tensor], top_k: int = -100, top_p: float = 100,
                                early_stop_num: int = -1, temperature: float = 1.0,
                                repetition_penalty: float = 1.35, noise: Optional[torch.Tensor] = None,
                                eos_mask_steps: int = 1, max_steps: int = 1500, trace: Optional[dict] = None,
                                **kwargs):
        """Returns (y_list, idx_list) exactly like reference t2s_model.py:583-779.
        noise: optional [steps, B, V] Exp(1) draws.  trace: dict filled with per-step
        logits (row-major [step][B, V]) when given."""
        B = len(x)
        P = prompts.shape[1]
        rows = []
        for b in range(B):
            xe = self.embed_text(x[b].long(), bert_feature[b])
            ye = self.embed_audio(prompts[b].long(), 0)
            rows.append(dict(X=xe.shape[0], xy=torch.cat([xe, ye], 0)))
        smax = max(r["X"] for r in rows) + P + max_steps + 1
        y = [prompts[b].long().clone() for b in range(B)]
        live = [True] * B
        y_list: List[Optional[torch.Tensor]] = [None] * B
        idx_list: List[Optional[int]] = [None] * B
        hs = [None] * B
        for r in rows:
            h, ks, vs = self.prefill_one(r["xy"], r["X"])
            r["n"] = r["xy"].shape[0]
            r["kc"] = [torch.zeros(min(smax, r["n"] + max_steps + 1), self.d, dtype=self.dtype) for _ in ks]
            r["vc"] = [torch.zeros_like(c) for c in r["kc"]]
            for li in range(self.L):
                r["kc"][li][: r["n"]] = ks[li]
                r["vc"][li][: r["n"]] = vs[li]
            r["h"] = h[-1:]
        if trace is not None:
            trace["logits"] = []
            trace["samples"] = []
        for idx in range(max_steps):
            act = [b for b in range(B) if live[b]]
            lg = torch.cat([self.logits(rows[b]["h"]) for b in act], 0)
            if idx < eos_mask_steps:
                lg = lg[:, :-1]
            prev = torch.stack([y[b] for b in act], 0)
            nz = None if noise is None else noise[idx][act]
            if trace is not None:
                full = torch.full((B, lg.shape[1]), float("nan"))
                full[act] = lg
                trace["logits"].append(full)
            samples, _ = sample(lg, prev, noise=nz, top_k=top_k, top_p=top_p,
                                repetition_penalty=repetition_penalty, temperature=temperature)
            # argmax of the repetition-penalised logits (in-place side effect in the reference)
            tokens = torch.argmax(apply_repetition_penalty(lg, prev, repetition_penalty), dim=-1)
            for j, b in enumerate(act):
                y[b] = torch.cat([y[b], samples[j].long()], 0)
            if trace is not None:
                s_full = torch.full((B,), -1, dtype=torch.long)
                s_full[act] = samples[:, 0].long()
                trace["samples"].append(s_full)
            for j, b in enumerate(act):
                if samples[j, 0].item() == self.EOS or tokens[j].item() == self.EOS:
                    live[b] = False
                    idx_list[b] = idx
                    y_list[b] = y[b][:-1]
            stop = False
            anyrow = next((b for b in range(B) if live[b]), None)
            gen = (y[anyrow].shape[0] - P) if anyrow is not None else None
            if (early_stop_num != -1 and gen is not None and gen > early_stop_num) or idx == max_steps - 1:
                stop = True
                for b in range(B):
                    if live[b]:
                        idx_list[b] = idx
                        y_list[b] = y[b][:-1]
            if None not in idx_list:
                stop = True
            if stop:
                break
            for b in range(B):
                if not live[b]:
                    continue
                r = rows[b]
                xin = self.embed_audio(y[b][-1:], P + idx)
                r["h"] = self.decode_one(xin, r["kc"], r["vc"], r["n"])
                r["n"] += 1
        return y_list, idx_list

    # -- H1b: naive single-sequence loop ----------------------------------------
    @torch.no_grad()
    def infer_panel_naive(self, x: torch.Tensor, x_lens, prompts: Optional[torch.Tensor], bert_feature: torch.Tensor,
                          top_k: int = -100, top_p: float = 100, early_stop_num: int = -1,
                          temperature: float = 1.0, repetition_penalty: float = 1.35,
                          noise: Optional[torch.Tensor] = None, **kwargs):
        """x [1, X], prompts [1, P], bert [1, 1024, X]; EOS masked while idx < 11
        (t2s_model.py:888-889); returns (y[:, :-1], idx) (t2s_model.py:916-918)."""
        ref_free = prompts is None            # t2s_model.py:849-856: y starts empty, positions start at 0, idx reported as 0
        if ref_free:
            prompts = torch.zeros(1, 0, dtype=torch.long)
        ys, idxs = self.infer_panel_batch_infer([x[0]], None, prompts, [bert_feature[0]], top_k=top_k, top_p=top_p,
                                                early_stop_num=early_stop_num, temperature=temperature,
                                                repetition_penalty=repetition_penalty, noise=noise,
                                                eos_mask_steps=11, **kwargs)
        return ys[0].unsqueeze(0), (0 if ref_free else idxs[0])
