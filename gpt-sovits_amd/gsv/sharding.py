"""Utterance sharding across the GPUs of one node.

The reference has no inference-side distribution.  After text segmentation every sentence is an
independent job sharing one prompt, so the path shards by utterance: weights and the prompt cache
are replicated, rank 0 scatters the tokenised segments, every rank runs AR -> decode locally with
no steady-state communication, and rank 0 gathers the int16 fragments and restores submission
order.  One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Payloads are KB (ids) and ~256 KB per 4 s utterance (int16), so plain
point-to-point collectives are used: one broadcast out, one length all-gather + one padded gather in.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from gsv.hostcopy import to_host


def deal_contiguous(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-sort the items (stable) and give each rank one contiguous run of the sorted order, so a
    rank's batches stay length-homogeneous (what to_batch's bucketing wants, reference
    TTS_infer_pack/TTS.py:859-879).  Runs differ in size by at most one item."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i])
    n = len(order)
    out, pos = [], 0
    for r in range(world):
        cnt = n // world + (1 if r < n % world else 0)
        out.append(order[pos:pos + cnt])
        pos += cnt
    return out


def pack_segments(segments: List[dict]) -> torch.Tensor:
    """int32 wire format: [n, len_0 .. len_{n-1}, text_len_0 .. text_len_{n-1}, phones...].
    BERT features are not shipped: the sharded path serves the non-zh case (all-zero features,
    reference TextPreprocessor.py:216-220); zh callers shard after their own BERT pass."""
    n = len(segments)
    lens = [len(s["phones"]) for s in segments]
    tl = [len(s["norm_text"]) for s in segments]
    flat = [p for s in segments for p in s["phones"]]
    return torch.tensor([n] + lens + tl + flat, dtype=torch.int32)


def unpack_segments(buf: torch.Tensor) -> List[dict]:
    a = buf.cpu().tolist()
    n = a[0]
    lens, tl = a[1:1 + n], a[1 + n:1 + 2 * n]
    out, o = [], 1 + 2 * n
    for i in range(n):
        ph = a[o:o + lens[i]]
        o += lens[i]
        # all-zero BERT features by definition of this wire format: None is the engine's spelling of that (no 1024 x X
        # zero block is allocated and scanned per segment and step)
        out.append({"phones": ph, "bert_features": None, "norm_text": "x" * tl[i]})
    return out


class ShardedSynthesizer:
    """`synth(segments) -> (int16 1-D numpy array or device tensor, per-fragment sample counts in the
    order of `segments`)` is the local engine call (TTS wrapper in production, a stub in the gloo tests)."""

    def __init__(self, synth: Callable[[List[dict]], Tuple[torch.Tensor, List[int]]], device: torch.device,
                 group=None):
        self.synth = synth
        self.device = device
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1

    def run(self, segments: Optional[List[dict]]) -> Optional[np.ndarray]:
        """rank 0 passes all segments (others pass None); rank 0 returns the int16 audio of all
        fragments in submission order, the other ranks return None."""
        dev = self.device
        # ---- scatter (one broadcast of the packed batch; ranks slice their share)
        if self.world > 1:
            hdr = torch.zeros(1, dtype=torch.int64, device=dev)
            if self.rank == 0:
                wire = pack_segments(segments).to(dev)
                hdr[0] = wire.numel()
            dist.broadcast(hdr, 0, group=self.group)
            if self.rank != 0:
                wire = torch.empty(int(hdr.item()), dtype=torch.int32, device=dev)
            dist.broadcast(wire, 0, group=self.group)
            segments = unpack_segments(wire)
        shares = deal_contiguous([len(s["norm_text"]) for s in segments], self.world)
        mine = shares[self.rank]
        import time as _time
        _t0 = _time.perf_counter()
        audio, frag_lens = self.synth([segments[i] for i in mine]) if mine else (np.zeros(0, dtype=np.int16), [])
        self.last_synth_s = _time.perf_counter() - _t0
        if self.world == 1:
            a = to_host(audio) if torch.is_tensor(audio) else audio
            assert a.dtype == np.int16 and sum(frag_lens) == a.size
            return _reorder([a], [frag_lens], shares, len(segments))
        if not torch.is_tensor(audio):
            audio = torch.from_numpy(audio).to(dev)     # the gather works on device buffers (RCCL)
        assert audio.dtype == torch.int16 and sum(frag_lens) == audio.numel()
        # ---- gather (lengths, then one padded gather to rank 0)
        n_local = torch.tensor([audio.numel()], dtype=torch.int64, device=dev)
        all_n = [torch.zeros_like(n_local) for _ in range(self.world)]
        dist.all_gather(all_n, n_local, group=self.group)
        maxn = max(int(t.item()) for t in all_n)
        # int16 samples travel as raw bytes: neither RCCL nor gloo has a 16-bit integer type
        pad16 = torch.zeros(max(maxn, 1), dtype=torch.int16, device=dev)
        pad16[: audio.numel()] = audio
        pad = pad16.view(torch.uint8)
        maxf = max(len(s) for s in shares)
        fl = torch.zeros(max(maxf, 1), dtype=torch.int64, device=dev)
        if frag_lens:
            fl[: len(frag_lens)] = torch.tensor(frag_lens, dtype=torch.int64, device=dev)
        if self.rank == 0:
            bufs = [torch.zeros_like(pad) for _ in range(self.world)]
            fbufs = [torch.zeros_like(fl) for _ in range(self.world)]
            dist.gather(pad, bufs, dst=0, group=self.group)
            dist.gather(fl, fbufs, dst=0, group=self.group)
            # submission order is restored ON THE DEVICE (one concatenation of fragment views), then a single copy brings
            # the result to the host: with 8 ranks the host-side concatenate of 70 MB was ~10 ms of rank 0's step
            lens = [f[: len(s)].cpu().tolist() for f, s in zip(fbufs, shares)]
            views = [b.view(torch.int16) for b in bufs]
            frags: List[Optional[torch.Tensor]] = [None] * len(segments)
            for v, ln_r, idxs in zip(views, lens, shares):
                o = 0
                for ln, i in zip(ln_r, idxs):
                    frags[i] = v[o:o + ln]
                    o += ln
            keep = [f for f in frags if f is not None]
            if not keep:
                return np.zeros(0, dtype=np.int16)
            return to_host(torch.cat(keep))
        dist.gather(pad, None, dst=0, group=self.group)
        dist.gather(fl, None, dst=0, group=self.group)
        return None


def _reorder(arrays: List[np.ndarray], frag_lens: List[List[int]], shares: List[List[int]], n: int) -> np.ndarray:
    frags: List[Optional[np.ndarray]] = [None] * n
    for arr, lens, idxs in zip(arrays, frag_lens, shares):
        o = 0
        for ln, i in zip(lens, idxs):
            frags[i] = arr[o:o + ln]
            o += ln
    return np.concatenate([f for f in frags if f is not None]) if n else np.zeros(0, dtype=np.int16)
