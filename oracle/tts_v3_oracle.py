"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the v3/v4 synthesis glue (H17): `TTS.using_vocoder_synthesis`
(reference TTS_infer_pack/TTS.py:1431-1494), `using_vocoder_synthesis_batched_infer` (:1496-1609) and
`sola_algorithm` (:1611-1637), written over three callables (decode_encp / cfm_inference / vocoder) so the same chain
runs over the oracles of those stages (each pinned against its reference class by oracle/gen_golden*.py).

PARITY UNPINNED for the glue itself: the reference's TTS module cannot be imported in the build container
(pytorch_lightning, ffmpeg, peft, ... are absent), so this file follows the reference's text line by line in
behaviour only; the stages it calls are pinned.  Never imported by the product path.
"""
import math
from typing import Callable, List

import torch
import torch.nn.functional as F

SPEC_MIN, SPEC_MAX = -12, 2     # TTS.py:55-56


def norm_spec(x):
    return (x - SPEC_MIN) / (SPEC_MAX - SPEC_MIN) * 2 - 1


def denorm_spec(x):
    return (x + 1) / 2 * (SPEC_MAX - SPEC_MIN) + SPEC_MIN


def sola_algorithm(frags: List[torch.Tensor], ov: int) -> torch.Tensor:
    frags = [f.clone() for f in frags]
    for i in range(len(frags) - 1):
        a, b = frags[i], frags[i + 1]
        tail, head = a[-ov:], b[:ov]
        corr = F.conv1d(tail.view(1, 1, -1), head.view(1, 1, -1), padding=ov // 2).view(-1)[:-1]
        idx = int(corr.argmax())
        n = ov - idx
        frags[i] = a[:-n]
        nb = b[idx:]
        win = torch.hann_window(2 * n, dtype=a.dtype)
        nb[:n] = win[:n] * nb[:n] + win[n:] * a[-n:]
        frags[i + 1] = nb
    return torch.cat(frags, 0)


def _prompt_features(decode_encp, prompt_semantic, prompt_phones, refer_spec, ref_mel, T_ref):
    fea_ref, ge = decode_encp(prompt_semantic.view(1, 1, -1), prompt_phones.view(1, -1), refer_spec, None, 1)
    mel2 = norm_spec(ref_mel)
    T_min = min(mel2.shape[2], fea_ref.shape[2])
    mel2, fea_ref = mel2[:, :, :T_min], fea_ref[:, :, :T_min]
    if T_min > T_ref:
        mel2, fea_ref, T_min = mel2[:, :, -T_ref:], fea_ref[:, :, -T_ref:], T_ref
    return fea_ref, ge, mel2, T_min


def using_vocoder_synthesis(decode_encp: Callable, cfm_inference: Callable, vocoder: Callable, vc: dict, prompt_semantic,
                            prompt_phones, refer_spec, ref_mel, semantic_tokens, phones, speed=1.0, sample_steps=32):
    """cfm_inference(fea [1,T,512], mel2 [1,100,Tp], n_steps, call_index) -> [1,100,T]"""
    fea_ref, ge, mel2, T_min = _prompt_features(decode_encp, prompt_semantic, prompt_phones, refer_spec, ref_mel, vc["T_ref"])
    chunk_len = vc["T_chunk"] - T_min
    fea_todo, ge = decode_encp(semantic_tokens, phones, refer_spec, ge, speed)
    outs, pos, call = [], 0, 0
    while True:
        chunk = fea_todo[:, :, pos:pos + chunk_len]
        if chunk.shape[-1] == 0:
            break
        pos += chunk_len
        fea = torch.cat([fea_ref, chunk], 2).transpose(2, 1)
        res = cfm_inference(fea, mel2, sample_steps, call)[:, :, mel2.shape[2]:]
        call += 1
        mel2 = res[:, :, -T_min:]
        fea_ref = chunk[:, :, -T_min:]
        outs.append(res)
    return vocoder(denorm_spec(torch.cat(outs, 2)))[0][0]


def using_vocoder_synthesis_batched_infer(decode_encp: Callable, cfm_inference: Callable, vocoder: Callable, vc: dict,
                                          prompt_semantic, prompt_phones, refer_spec, ref_mel, idx_list, semantic_tokens_list,
                                          batch_phones, speed=1.0, sample_steps=32, sola=sola_algorithm):
    fea_ref, ge, mel2, T_min = _prompt_features(decode_encp, prompt_semantic, prompt_phones, refer_spec, ref_mel, vc["T_ref"])
    chunk_len = vc["T_chunk"] - T_min
    ov, up = vc["overlapped_len"], vc["upsample_rate"]
    feats, lens = [], []
    for i, idx in enumerate(idx_list):
        f, _ = decode_encp(semantic_tokens_list[i][-idx:].view(1, 1, -1), batch_phones[i].view(1, -1), refer_spec, ge, speed)
        feats.append(f)
        lens.append(f.shape[2])
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
    spec = cfm_inference(fea, mel2, sample_steps, 0)[:, :, -chunk_len:]
    spec = spec.permute(1, 0, 2).contiguous().view(spec.shape[1], -1).unsqueeze(0)
    audio = vocoder(denorm_spec(spec))[0][0]
    pieces, p = [], 0
    while p < audio.shape[-1]:
        pieces.append(audio[p:p + chunk_len * up])
        p += chunk_len * up
    audio = sola(pieces, ov * up)
    audio = audio[ov * up:-pad_len * up]           # as written in the reference: empty when pad_len == 0 (TTS.py:1600)
    out = []
    for n in lens:
        out.append(audio[:n * up])
        audio = audio[n * up:]
    return out
