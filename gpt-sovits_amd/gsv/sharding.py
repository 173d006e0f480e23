"""Utterance sharding across the GPUs of one node (BASELINE configs[2] and [4]).

The reference has no inference-side distribution.  After text segmentation every sentence is an independent job sharing one
prompt, so the path shards by utterance: weights and the prompt cache are replicated, rank 0 broadcasts the tokenised
segments (KB; BERT blocks only for zh text), the length-sorted segments are cut into BATCHES of `batch_size`, every rank
synthesises whole batches locally with no steady-state communication, and each finished batch travels to rank 0 as int16
fragments, point to point (xGMI: one link per peer into rank 0, no ring, no all-reduce).

* Work queue: batches are handed out through an atomic counter in the process group's store (`store.add`), so a rank whose
  rows hit EOS early simply takes the next batch ("next bucket goes to the first idle GPU"); without a store the
  batches are dealt round-robin.
* Streaming (configs[4], reference `return_fragment`, TTS.py:1321): `run_stream` yields on rank 0 as soon as the next batch
  IN ORDER has arrived; rank 0 works on its own batches in between and only blocks on a receive when the queue is empty.
* `run` (configs[1] / [2]) returns the whole job's audio in submission order; every rank keeps its int16 results on its GPU
  and sends ONE record to rank 0 over RCCL when its queue is empty (no send is ever in flight beside a decode engine).
* Streamed fragments (`run_stream`) travel over a host-side gloo group: they are int16 on the host already, and an RCCL send
  posted before rank 0's receive would keep a kernel resident on the sender's GPU while its persistent decode engine needs
  every CU.
* A rank whose synthesis raises still sends its record (with an error flag) so that nobody waits for it; rank 0 receives
  every outstanding record -- also when its own batch failed -- and re-raises afterwards.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import time
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from gsv.hostcopy import to_host


def deal_contiguous(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-sort the items (stable) and give each rank one contiguous run of the sorted order; runs differ in size by at
    most one item (kept for callers that want one share per rank)."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i])
    n = len(order)
    out, pos = [], 0
    for r in range(world):
        cnt = n // world + (1 if r < n % world else 0)
        out.append(order[pos:pos + cnt])
        pos += cnt
    return out


def make_batches(lengths: Sequence[int], batch_size: int, bucket: bool = True, first_batch: int = 0) -> List[List[int]]:
    """Length-sorted (stable) order cut into runs of `batch_size`: batches stay length-homogeneous, which is what
    to_batch's bucketing wants (reference TTS_infer_pack/TTS.py:859-879).  bucket=False: submission order cut into runs --
    what the reference does when fragments are streamed (`return_fragment` switches bucketing off, TTS.py:1050-1054), so
    that a long text is heard in reading order."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i]) if bucket else list(range(len(lengths)))
    head = min(first_batch, len(order)) if first_batch and 0 < first_batch < batch_size else 0
    # `first_batch` (streaming): a smaller first batch is heard sooner -- the AR decode is latency-bound, so 8 sentences cost
    # almost the same decode time as 32 but a quarter of the SoVITS pass and of the prefill
    out = [order[:head]] if head else []
    return out + [order[i:i + batch_size] for i in range(head, len(order), batch_size)]


def _nonzero(b) -> bool:
    """any(b != 0); host data is tested through numpy (a torch reduction per segment costs ~1 ms on a many-core host)"""
    t = torch.as_tensor(b)
    if t.device.type == "cpu":
        try:
            return bool(np.any(t.detach().numpy()))
        except (TypeError, RuntimeError):
            pass
    return bool(t.any())


def pack_segments(segments: List[dict], ship_bert: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """-> (int32 wire [n, has_bert_0.., len_0.., text_len_0.., phones...], float16 BERT blob [1024, sum of shipped columns] or
    None).  All-zero BERT features (every non-zh segment, reference TextPreprocessor.py:216-220) are never shipped: None on the
    receiving side is the engine's spelling of zeros.  Non-zero features are shipped, or refused loudly with ship_bert=False."""
    n = len(segments)
    lens = [len(s["phones"]) for s in segments]
    tl = [len(s["norm_text"]) for s in segments]
    has, blobs = [], []
    for s in segments:
        b = s.get("bert_features")
        nz = b is not None and _nonzero(b)
        if nz and not ship_bert:
            raise ValueError("pack_segments: a segment carries non-zero BERT features (zh text) and ship_bert=False would "
                             "synthesise it as if they were zero")
        if nz and tuple(b.shape) != (1024, len(s["phones"])):
            raise ValueError(f"bert_features must be [1024, {len(s['phones'])}], got {tuple(b.shape)}")
        has.append(int(nz))
        if nz:
            blobs.append(torch.as_tensor(b).to(torch.float16).cpu())
    flat = [p for s in segments for p in s["phones"]]
    wire = torch.tensor([n] + has + lens + tl + flat, dtype=torch.int32)
    return wire, (torch.cat(blobs, 1).contiguous() if blobs else None)


def unpack_segments(wire: torch.Tensor, bert: Optional[torch.Tensor] = None) -> List[dict]:
    a = wire.cpu().tolist()
    n = a[0]
    has, lens, tl = a[1:1 + n], a[1 + n:1 + 2 * n], a[1 + 2 * n:1 + 3 * n]
    out, o, bo = [], 1 + 3 * n, 0
    for i in range(n):
        ph = a[o:o + lens[i]]
        o += lens[i]
        bf = None
        if has[i]:
            bf = bert[:, bo:bo + lens[i]].float().cpu()
            bo += lens[i]
        out.append({"phones": ph, "bert_features": bf, "norm_text": "x" * tl[i]})
    return out


def join_fragments(frags: List[np.ndarray]) -> np.ndarray:
    """int16 fragments in submission order -> one array.  When the fragments already lie back to back in ONE host buffer
    (a job of one batch whose order was not permuted: BASELINE configs[1]) the result is a view of that buffer; otherwise a
    concatenation.  The copy it saves is 8 MB per batch of 32 x 4 s, ~1 ms of host time per step."""
    if not frags:
        return np.zeros(0, dtype=np.int16)
    frags = [f for f in frags if f.size] or frags[:1]         # empty fragments add nothing (and carry no address)
    base = frags[0].base
    if isinstance(base, np.ndarray) and base.ndim == 1 and base.dtype == np.int16 and base.flags.c_contiguous:
        addr = base.__array_interface__["data"][0]
        start = frags[0].__array_interface__["data"][0]
        nxt, ok = start, True
        for f in frags:
            if f.base is not base or f.ndim != 1 or f.dtype != np.int16 or (f.size > 1 and f.strides[0] != 2) \
                    or f.__array_interface__["data"][0] != nxt:
                ok = False
                break
            nxt += 2 * f.size
        if ok and (start - addr) % 2 == 0:
            o = (start - addr) // 2
            return base[o:o + (nxt - start) // 2]
    return np.concatenate(frags)


def place_pieces(pieces, batches: List[List[int]], n_segments: int, dev: torch.device) -> np.ndarray:
    """`pieces` = (batch index, int16 source tensor, offset of the batch's samples in it, fragment lengths in the batch's
    order), sources on the host or on `dev`.  Returns the audio of all segments in submission order: ONE buffer (page-locked
    when `dev` is a GPU), every piece copied once, straight to its place -- a batch of consecutive segments as one copy (D2H
    for the other ranks' records, host to host for rank 0's own).  Before: one D2H of the whole record + a concatenation of
    everything (66 MB at N = 8).  A batch nobody delivered leaves its segments empty."""
    seg_len = [0] * n_segments
    for k, _src, _o, fl in pieces:
        for i, ln in zip(batches[k], fl):
            seg_len[i] = int(ln)
    offs = np.concatenate([[0], np.cumsum(seg_len, dtype=np.int64)])
    cuda = dev.type == "cuda"
    result = torch.empty(int(offs[-1]), dtype=torch.int16, pin_memory=cuda)
    for k, src, o, fl in pieces:
        idx = batches[k]
        if all(idx[j] + 1 == idx[j + 1] for j in range(len(idx) - 1)):
            n = int(sum(fl))
            result[int(offs[idx[0]]):int(offs[idx[0]]) + n].copy_(src[o:o + n], non_blocking=True)
        else:
            for i, ln in zip(idx, fl):
                result[int(offs[i]):int(offs[i]) + int(ln)].copy_(src[o:o + int(ln)], non_blocking=True)
                o += int(ln)
    if cuda:
        torch.cuda.current_stream(dev).synchronize()
    return result.numpy()


class ShardedSynthesizer:
    """`synth(segments) -> (int16 1-D numpy array or device tensor, per-fragment sample counts in the order of `segments`)`
    is the local engine call (the TTS wrapper in production, a stub in the gloo tests)."""

    def __init__(self, synth: Callable[[List[dict]], Tuple[torch.Tensor, List[int]]], device: torch.device, group=None,
                 dynamic: bool = True):
        self.synth = synth
        self.device = device
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.store = None
        if self.on and dynamic and self.world > 1:
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                self.store = _get_default_store()
            except Exception:
                self.store = None
        self._job = 0
        self._hg = None
        self.last_synth_s = 0.0
        self.last_owner: List[int] = []

    # ---- scatter ---------------------------------------------------------------------------------------------------
    def _broadcast_segments(self, segments: Optional[List[dict]], batch_size: int, ship_bert: bool):
        if self.world == 1:
            return segments, batch_size
        dev = self.device
        hdr = torch.zeros(3, dtype=torch.int64, device=dev)
        wire = bert = None
        if self.rank == 0:
            wire, bert = pack_segments(segments, ship_bert)
            hdr[0], hdr[1], hdr[2] = wire.numel(), (0 if bert is None else bert.shape[1]), batch_size
        dist.broadcast(hdr, 0, group=self.group)
        nw, nb, batch_size = (int(v) for v in hdr.tolist())
        wire = wire.to(dev) if self.rank == 0 else torch.empty(nw, dtype=torch.int32, device=dev)
        dist.broadcast(wire, 0, group=self.group)
        if nb:
            bert = bert.to(dev) if self.rank == 0 else torch.empty(1024, nb, dtype=torch.float16, device=dev)
            dist.broadcast(bert, 0, group=self.group)
        if self.rank != 0:
            segments = unpack_segments(wire, bert if nb else None)
        return segments, batch_size

    # ---- work queue ------------------------------------------------------------------------------------------------
    def _next_batch(self, job: int, nb: int, taken: List[int]) -> Optional[int]:
        """index of the next unclaimed batch, or None.  Dynamic: atomic counter in the store; static: round-robin."""
        if self.world == 1:
            k = len(taken)
            return k if k < nb else None
        if self.store is not None:
            k = int(self.store.add(f"gsv_shard_next_{job}", 1)) - 1
            if k >= nb:
                return None
            self.store.set(f"gsv_shard_owner_{job}_{k}", str(self.rank))
            return k
        k = self.rank + self.world * len(taken)
        return k if k < nb else None

    def _owner(self, job: int, k: int) -> int:
        if self.store is not None:
            return int(self.store.get(f"gsv_shard_owner_{job}_{k}"))          # blocks until the claimer has set it
        return k % self.world

    def _synth_batch(self, segs: List[dict]):
        t0 = time.perf_counter()
        try:
            audio, frag_lens = self.synth(segs)
            err = 0
        except Exception as e:                                                 # noqa: BLE001 -- reported to rank 0, re-raised there
            self._last_exc = e
            audio, frag_lens, err = np.zeros(0, dtype=np.int16), [0] * len(segs), 1
        self.last_synth_s += time.perf_counter() - t0
        if not torch.is_tensor(audio):
            audio = torch.from_numpy(np.ascontiguousarray(audio))
        assert audio.dtype == torch.int16 and sum(frag_lens) == audio.numel() and len(frag_lens) == len(segs)
        return audio, list(frag_lens), err

    # ---- the job ---------------------------------------------------------------------------------------------------
    def _host_group(self):
        """process group for HOST-side point-to-point traffic (streamed fragments).  Under RCCL a posted send is a kernel
        that stays resident on the sender's GPU until rank 0 posts the matching receive; the persistent AR decode engine
        needs every CU, so a send left in flight while the next batch decodes can starve one of its workgroups
        (ADVICE r2).  Streamed fragments are int16 on the host already, so they travel over a gloo group instead; the
        job-level gather of `run` uses RCCL at a point where no engine is running."""
        if self._hg is None:
            if not self.on or self.world == 1 or dist.get_backend(self.group) == "gloo":
                self._hg = self.group
            else:
                self._hg = dist.new_group(backend="gloo")          # collective: every rank constructs it at the same point
        return self._hg

    def run_stream(self, segments: Optional[List[dict]], batch_size: int = 32, ship_bert: bool = True, bucket: bool = True,
                   first_batch: int = 0) -> Iterator[Tuple[List[int], List[np.ndarray]]]:
        """rank 0 passes all segments (the others None).  On rank 0 yields (segment indices of the batch, their int16 fragments)
        batch by batch in the order of `make_batches` (bucket=False: submission order, BASELINE configs[4]'s streamed long
        text); on the other ranks yields nothing but must be iterated to the end.
        A failure anywhere (rank 0's own batch included) is raised on rank 0 AFTER every outstanding batch of the other
        ranks has been received, so that no rank is left waiting in a send."""
        self._job += 1
        job = self._job
        self.last_synth_s = 0.0
        hg = self._host_group() if self.world > 1 else None
        # the sign carries `bucket`, the bits above 20 `first_batch` to the other ranks (one broadcast header)
        code = (batch_size + (max(0, int(first_batch)) << 20)) * (1 if bucket else -1)
        segments, code = self._broadcast_segments(segments, code, ship_bert)
        bucket, code = code > 0, abs(code)
        batch_size, first_batch = code & ((1 << 20) - 1), code >> 20
        batches = make_batches([len(s["norm_text"]) for s in segments], batch_size, bucket, first_batch)
        nb = len(batches)
        done = {}                       # rank 0: batch index -> (audio host array, frag_lens) or None (failed)
        taken: List[int] = []
        pending = []                    # other ranks: in-flight sends (buffers must outlive them)
        self.last_owner = [-1] * nb
        failure: List[Optional[BaseException]] = [None]

        def work_one() -> bool:
            k = self._next_batch(job, nb, taken)
            if k is None:
                return False
            taken.append(k)
            audio, frag_lens, err = self._synth_batch([segments[i] for i in batches[k]])
            host = to_host(audio) if audio.is_cuda else audio.numpy()
            if self.rank == 0:
                if err:
                    done[k] = None
                    failure[0] = failure[0] or self._last_exc
                else:
                    done[k] = (host, frag_lens)
            else:
                # header: [batch, error, n_samples, frag lens...] then the samples as raw bytes (gloo has no int16)
                hdr = torch.tensor([k, err, int(host.size)] + frag_lens, dtype=torch.int64)
                pay = torch.from_numpy(np.ascontiguousarray(host)).view(torch.uint8) if host.size else torch.zeros(2, dtype=torch.uint8)
                pending.append((hdr, pay, dist.isend(hdr, 0, group=hg, tag=2 * k), dist.isend(pay, 0, group=hg, tag=2 * k + 1)))
            return True

        if self.rank != 0:
            while work_one():
                pass
            for _, _, h1, h2 in pending:
                h1.wait(); h2.wait()
            return
        # rank 0: emit in order; do own work whenever the next batch in order is not there yet
        more = True
        for k in range(nb):
            while k not in done:
                if more and failure[0] is None:
                    more = work_one()
                    continue
                if self.world == 1 or (self.store is None and k % self.world == 0):
                    done[k] = None          # rank 0's own batch that will not be synthesised any more (a failure came first)
                    continue
                owner = self._owner(job, k)
                assert owner != 0, "a batch claimed by rank 0 is always in `done`"
                hdr = torch.zeros(3 + len(batches[k]), dtype=torch.int64)
                dist.recv(hdr, owner, group=hg, tag=2 * k)
                h = hdr.tolist()
                pay = torch.zeros(max(2 * h[2], 2), dtype=torch.uint8)
                dist.recv(pay, owner, group=hg, tag=2 * k + 1)
                self.last_owner[k] = owner
                if h[1]:
                    failure[0] = failure[0] or RuntimeError(f"rank {owner} failed while synthesising batch {k}")
                    done[k] = None
                else:
                    done[k] = (pay[: 2 * h[2]].view(torch.int16).numpy().copy(), [int(v) for v in h[3:]])
            if self.last_owner[k] < 0:
                self.last_owner[k] = 0
            item = done.pop(k)
            if item is None or failure[0] is not None:
                continue                    # after a failure: keep receiving (drain), stop emitting
            audio, frag_lens = item
            frags, o = [], 0
            for ln in frag_lens:
                frags.append(audio[o:o + ln])
                o += ln
            yield list(batches[k]), frags
        if failure[0] is not None:
            raise failure[0]

    def run(self, segments: Optional[List[dict]], batch_size: Optional[int] = None, ship_bert: bool = True) -> Optional[np.ndarray]:
        """whole job: rank 0 returns the int16 audio of all fragments in submission order, the other ranks None.
        `batch_size` None = one batch per rank (configs[1]'s weak-scaling step: 32 utterances per GPU).

        Every rank synthesises the batches it claims and keeps the int16 result on its GPU; when its queue is empty it sends
        ONE record to rank 0 over the device group (RCCL point to point over xGMI: sizes, then fragment table, then samples),
        i.e. the gather happens when no decode engine is running on the sender and rank 0 has finished its own work."""
        if batch_size is None:
            n = len(segments) if self.rank == 0 else 0
            if self.world > 1:
                t = torch.tensor([n], dtype=torch.int64, device=self.device)
                dist.broadcast(t, 0, group=self.group)
                n = int(t.item())
            batch_size = max(1, -(-n // self.world))
        if self.world == 1:
            n_total = len(segments)
            out: List[Optional[np.ndarray]] = [None] * n_total
            for idxs, frags in self.run_stream(segments, batch_size, ship_bert):
                for i, f in zip(idxs, frags):
                    out[i] = f
            keep = [f for f in out if f is not None]
            return join_fragments(keep)
        dev = self.device
        self._job += 1
        job = self._job
        self.last_synth_s = 0.0
        segments, batch_size = self._broadcast_segments(segments, batch_size, ship_bert)
        batches = make_batches([len(s["norm_text"]) for s in segments], batch_size)
        nb = len(batches)
        self.last_owner = [-1] * nb
        taken: List[int] = []
        mine = []                           # (batch, audio tensor, frag_lens)
        err, exc = 0, None
        while True:
            k = self._next_batch(job, nb, taken)
            if k is None:
                break
            taken.append(k)
            audio, frag_lens, e = self._synth_batch([segments[i] for i in batches[k]])
            if e:
                err, exc = 1, self._last_exc
                break                       # stop claiming: with the work queue the other ranks take what is left
            mine.append((k, audio, frag_lens))
        table = [v for k, _a, fl in mine for v in [k, len(fl)] + fl]
        if self.rank != 0:
            pay = torch.cat([a.to(dev).reshape(-1) for _k, a, _f in mine]) if mine else torch.zeros(1, dtype=torch.int16, device=dev)
            sizes = torch.tensor([len(table), int(pay.numel()), err], dtype=torch.int64, device=dev)
            dist.send(sizes, 0, group=self.group)
            dist.send(torch.tensor(table + [0], dtype=torch.int64, device=dev), 0, group=self.group)
            dist.send(pay.view(torch.uint8), 0, group=self.group)
            return None
        failed = [f"rank 0: {exc!r}"] if err else []
        # sizes and fragment tables of every rank first (the receives are stream-ordered; the payloads follow into ONE device
        # buffer: 7 x 8 MB of int16 at N = 8), so that every fragment's place in the result is known before a sample moves
        recs = []
        for r in range(1, self.world):
            sizes = torch.zeros(3, dtype=torch.int64, device=dev)
            dist.recv(sizes, r, group=self.group)
            nt, ns, rerr = (int(v) for v in sizes.tolist())
            tab = torch.zeros(nt + 1, dtype=torch.int64, device=dev)
            dist.recv(tab, r, group=self.group)
            recs.append((r, nt, max(ns, 1), rerr, tab))
        big = torch.zeros(2 * sum(rec[2] for rec in recs), dtype=torch.uint8, device=dev)
        o8 = 0
        for r, nt, ns, rerr, tab in recs:
            dist.recv(big[o8:o8 + 2 * ns], r, group=self.group)
            o8 += 2 * ns
        big16 = big.view(torch.int16)
        # pieces = (batch, source tensor, offset in it, fragment lengths); a batch nobody delivered leaves its segments empty
        pieces = []
        for k, a, fl in mine:
            self.last_owner[k] = 0
            pieces.append((k, a.reshape(-1), 0, fl))
        base = 0
        for r, nt, ns, rerr, tab in recs:
            if rerr:
                failed.append(f"rank {r}")
            t, p, o = tab.tolist(), 0, base
            while p < nt:
                k, nf = t[p], t[p + 1]
                fl = t[p + 2:p + 2 + nf]
                self.last_owner[k] = r
                pieces.append((k, big16, o, fl))
                o += sum(fl)
                p += 2 + nf
            base += ns
        if failed:
            if exc is not None:
                raise exc
            raise RuntimeError(f"sharded synthesis failed on {', '.join(failed)}")
        return place_pieces(pieces, batches, len(segments), dev)
