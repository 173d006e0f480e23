"""GPU: the persistent AR decode engine (csrc/t2s_mega.hip) against the launch-per-phase step of the same library, against
the fp32 engine (whose ids are bit-exact vs the reference goldens, test_t2s_gpu.py) and against the reference goldens.

The persistent engine is fp16 (production dtype).  fp16 ids cannot be bit-exact vs the fp32 reference (greedy argmax flips
on near-ties), so the bars are: (1) one full pass (24 layers + logits) from identical inputs gives logits within 3e-2 of
the launch path's fp16 logits and 6e-2 of the fp32 engine's; (2) free-running ids equal the reference golden while the
golden top-2 margin is clear (same bar as the launch path); (3) engine vs launch path over the BASELINE config-2 workload:
rows identical or diverging only after a long common prefix, agreement rate printed; (4) ragged batches with EOS finishes,
every batch size 1..32, injected-noise sampling: same finishing bookkeeping as the launch path.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(cfg, sd, dtype=torch.float16, max_batch=32, max_seq=512):
    from gsv.AR.models.t2s_model import Text2SemanticDecoder
    m = Text2SemanticDecoder(cfg, device=DEV, dtype=dtype, max_batch=max_batch, max_seq=max_seq)
    m.load_state_dict(sd)
    return m


def _v2(seed=0, suppress_eos=True):
    from gsv import synthetic as S
    cfg = S.T2S_V2_CONFIG
    return cfg, S.make_t2s_state_dict(cfg, seed=seed, suppress_eos=suppress_eos)


def _batch(n, seed=0):
    from gsv import synthetic as S
    utt = S.make_utterances(n, seed=seed)
    xs = [torch.tensor(it["all_phones"], device=DEV) for it in utt["items"]]
    berts = [it["bert"].to(DEV) for it in utt["items"]]
    prompts = utt["prompt_semantic"].unsqueeze(0).expand(n, -1).contiguous().to(DEV)
    return xs, berts, prompts


def _common_prefix(a, b):
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    return n


@pytest.fixture(scope="module")
def v2_engine():
    cfg, sd = _v2()
    return _engine(cfg, sd)


def test_engine_is_selected_and_reports_it(v2_engine):
    eng = v2_engine
    xs, berts, prompts = _batch(8)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=6, repetition_penalty=1.35)
    eng.set_mega(True)
    ys, idx = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    mode, ms, steps = eng.decode_info()
    assert mode == 1 and steps == 6 and ms > 0, "the persistent engine must run for fp16 / v2 shape / B <= 32 on an MI355X"
    assert idx == [6] * 8
    eng.set_mega(False)
    ys0, idx0 = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 0
    eng.set_mega(True)
    assert idx0 == idx


@pytest.mark.parametrize("B", [1, 3, 8, 32])
def test_one_pass_logits_match_launch_path_and_fp32(v2_engine, B):
    """early_stop_num = 1: step 0 (shared kernels) + ONE decode step through each path from identical state."""
    eng = v2_engine
    xs, berts, prompts = _batch(B)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=1, repetition_penalty=1.35)
    eng.set_mega(True)
    ya, _ = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 1
    la = eng.debug_logits(B).cpu().numpy()
    eng.set_mega(False)
    yb, _ = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    lb = eng.debug_logits(B).cpu().numpy()
    eng.set_mega(True)
    assert np.isfinite(la).all()
    err = np.abs(la - lb).max()
    print(f"[mega] B={B}: one-pass logits vs launch path max-abs {err:.3e} (|logits| max {np.abs(lb).max():.1f})")
    assert err < 3e-2
    cfg, sd = _v2()
    e32 = _engine(cfg, sd, dtype=torch.float32, max_batch=max(B, 1), max_seq=320)
    e32.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    l32 = e32.debug_logits(B).cpu().numpy()
    assert np.abs(la - l32).max() < 6e-2


def test_golden_v2_ids_on_clear_margins():
    """ragged 24-layer case with EOS bookkeeping: ids equal the reference golden while its top-2 margin > 0.25."""
    name = "t2s_v2_greedy"
    case = cases.T2S_CASES[name]
    cfg, sd, xs, berts, prompts, noise = cases.t2s_case_inputs(case)
    g = load_golden(name)
    eng = _engine(cfg, sd, max_batch=8, max_seq=256)
    ys, idxs = eng.infer_panel_batch_infer([x.to(DEV) for x in xs], None, prompts.to(DEV), [b.to(DEV) for b in berts],
                                           top_k=1, top_p=1.0, temperature=1.0, early_stop_num=case["early_stop"],
                                           repetition_penalty=case["rep"])
    assert eng.decode_info()[0] == 1
    margins = g["min_top2_margin"]
    o, gold = 0, []
    for n in g["y_lens"]:
        gold.append(g["y_flat"][o:o + n])
        o += n
    P = prompts.shape[1]
    for a, b in zip(ys, gold):
        a = a.cpu().tolist()
        n = 0
        while n < min(len(a), len(b)) - P and margins[n] > 0.25:
            n += 1
        assert a[: P + n] == b[: P + n].tolist()


def test_config2_workload_agreement_with_launch_path_and_fp32(v2_engine):
    """BASELINE configs[1]: B = 32 x (80 phonemes, 100 prompt tokens), 100 generated tokens, greedy.  Reports the
    free-running token agreement engine vs launch path (both fp16) and engine vs fp32 engine."""
    eng = v2_engine
    xs, berts, prompts = _batch(32)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=100, repetition_penalty=1.35)
    eng.set_mega(True)
    ya, ia = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 1
    ya2, _ = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert [t.tolist() for t in ya] == [t.tolist() for t in ya2], "the engine must be deterministic"
    eng.set_mega(False)
    yb, ib = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    eng.set_mega(True)
    assert ia == [100] * 32 and ib == ia
    cfg, sd = _v2()
    e32 = _engine(cfg, sd, dtype=torch.float32, max_batch=32, max_seq=320)
    yc, _ = e32.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    P = prompts.shape[1]
    pa = [_common_prefix(a.tolist()[P:], b.tolist()[P:]) for a, b in zip(ya, yb)]
    pc = [_common_prefix(a.tolist()[P:], c.tolist()[P:]) for a, c in zip(ya, yc)]
    pbc = [_common_prefix(b.tolist()[P:], c.tolist()[P:]) for b, c in zip(yb, yc)]
    print(f"[mega] config-2 free-running agreement over 32 x 100 tokens: engine vs launch path {sum(pa)}/3200 common-prefix "
          f"tokens ({sum(p == 100 for p in pa)}/32 rows identical); engine vs fp32 {sum(pc)}/3200 "
          f"({sum(p == 100 for p in pc)}/32 rows); launch path vs fp32 {sum(pbc)}/3200 ({sum(p == 100 for p in pbc)}/32 rows)")
    for y in ya:
        assert y.shape[0] == P + 100 and int(y.max()) < 1024 and int(y.min()) >= 0
    # the two fp16 paths share every GEMM's summation order; only attention and LayerNorm reduce in a different order
    assert sum(pa) >= 0.5 * 3200
    # against fp32 the engine must do no worse than the launch path does (same dtype, same rounding points)
    assert sum(pc) >= 0.8 * sum(pbc) - 100


@pytest.mark.parametrize("B", [1, 2, 5, 9, 17, 31])
def test_eos_finishes_ragged_rows_every_batch_size(B):
    """rows finish by EOS at different steps (the EOS row of ar_predict_layer is made a slightly amplified copy of a
    token that row 0 emits early, so EOS wins the argmax whenever that token would), ragged text lengths, every batch
    size: finishing bookkeeping (idx, lengths, token ranges) is self-consistent and, for rows whose ids agree with the
    launch path, identical to it; a row may differ from the launch path only from a step on at which the launch path's
    own top-2 margin is a near-tie (< 0.1; both paths carry <= 3e-2 of fp16 noise); rows of a group keep running after a
    neighbour has finished."""
    from gsv import synthetic as S
    cfg, sd = _v2(seed=3, suppress_eos=True)
    utt = S.make_utterances(B)
    g = torch.Generator().manual_seed(B)
    xs = []
    for it in utt["items"]:
        n = 20 + int(torch.randint(0, 60, (1,), generator=g))
        xs.append(torch.tensor(it["all_phones"][:n], device=DEV))
    berts = [it["bert"][:, : x.shape[0]].to(DEV) for it, x in zip(utt["items"], xs)]
    prompts = utt["prompt_semantic"].unsqueeze(0).expand(B, -1).contiguous().to(DEV)
    P = prompts.shape[1]
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=48, repetition_penalty=1.35)
    probe = _engine(cfg, sd, max_batch=32, max_seq=400)
    y0, _ = probe.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    # the trigger token: the one whose first occurrence (in the EOS-free run) is spread over the most different steps
    # across the rows, none of them in the first two steps
    gen = [y[P:].tolist() for y in y0]
    best, tstar = -1, None
    for tok in sorted(set(gen[0][2:])):
        firsts = [g.index(tok) if tok in g else 48 for g in gen]
        if min(firsts) < 2:
            continue
        score = len(set(firsts)) if B > 1 else 1
        if score > best:
            best, tstar = score, tok
    assert tstar is not None
    expect_first = [g.index(tstar) if tstar in g else 48 for g in gen]
    sd = dict(sd)
    w = sd["ar_predict_layer.weight"].clone()
    w[1024] = 1.002 * w[tstar]
    sd["ar_predict_layer.weight"] = w
    eng = _engine(cfg, sd, max_batch=32, max_seq=400)
    eng.set_mega(True)
    ya, ia = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 1
    eng.set_mega(False)
    yb, ib = eng.infer_panel_batch_infer(xs, None, prompts, berts, dump_logits=True, **kw)
    Lb = eng.last_logits_dump.cpu()            # launch path's raw logits [steps][B][V]
    eng.set_mega(True)
    from oracle.t2s_oracle import apply_repetition_penalty
    same, near_ties = 0, []
    for r, (a, b, na, nb) in enumerate(zip(ya, yb, ia, ib)):
        assert a.shape[0] == P + na and 0 <= na <= 48
        assert na == 0 or int(a[P:].max().item()) < 1024
        if a.tolist() == b.tolist() and na == nb:
            same += 1
            continue
        # The two fp16 paths may part ways only at a NEAR-TIE.  The case builds one on purpose: the EOS row of the predict layer
        # is 1.002 x the trigger token's row, so whenever the trigger token leads, EOS leads it by 0.2 % of the logit (~0.02 at
        # |logit| ~ 10) -- inside the fp16 noise band of either path (<= 3e-2 each, test_one_pass_logits...).  That is what made
        # B = 2 finish at [6, 7] vs [6, 6] in profiles/r02_parity_log.txt.  Asserted here: at the first step where the rows
        # differ (a different token, or one path finishing), the launch path's penalised top-2 margin is below 0.1.
        ga, gb = a.tolist()[P:], b.tolist()[P:]
        dstep = _common_prefix(ga, gb)
        lg = Lb[dstep, r].clone().unsqueeze(0)
        if dstep < 1:
            lg = lg[:, :-1]
        pen = apply_repetition_penalty(lg, b[:P + dstep].cpu().view(1, -1), 1.35)[0]
        top2 = torch.topk(pen, 2).values
        margin = float(top2[0] - top2[1])
        near_ties.append((r, dstep, round(margin, 4)))
        assert margin < 0.1, f"row {r} leaves the launch path at step {dstep} where its top-2 margin is {margin:.3f} (not a near-tie)"
    print(f"[mega] B={B}: {same}/{B} rows identical to the launch path; finish steps engine {ia} launch path {ib}; "
          f"divergences (row, step, launch-path top-2 margin): {near_ties}")
    assert min(ia) < 48 and (B < 9 or len(set(ia)) > 1), f"the case must contain EOS finishes at different steps: {ia} (expected about {expect_first})"


def test_direct_abi_refuses_budget_beyond_arena():
    """gsv_t2s_decode through ctypes with max_steps that does not fit the K/V arena returns an error code, not tokens."""
    import ctypes as C
    from gsv import _lib
    cfg, sd = _v2()
    eng = _engine(cfg, sd, max_batch=4, max_seq=256)
    xs, berts, prompts = _batch(2)
    phones = torch.cat(xs).to(DEV, torch.int32)
    lens = (C.c_int32 * 2)(*[int(x.shape[0]) for x in xs])
    pr = prompts.to(DEV, torch.int32).contiguous()
    P = int(pr.shape[1])
    s = C.c_void_p(eng.stream.cuda_stream)
    l = _lib.lib()
    _lib.check(l.gsv_t2s_prefill(eng._h, phones.data_ptr(), C.cast(lens, C.c_void_p), 2, None, pr.data_ptr(), P, s))
    room = 256 - (max(int(x.shape[0]) for x in xs) + P)
    out = torch.zeros(2, 1500, dtype=torch.int32, device=DEV)
    ol = torch.zeros(2, dtype=torch.int32, device=DEV)
    steps = C.c_int(0)
    sp = _lib.SamplingParams(1, 1.0, 1.0, 1.35, -1, 1, room + 1, 0)
    rc = l.gsv_t2s_decode(eng._h, C.byref(sp), None, 0, out.data_ptr(), ol.data_ptr(), C.byref(steps), s)
    assert rc == -1 and b"arena" in l.gsv_last_error().lower()
    sp = _lib.SamplingParams(1, 1.0, 1.0, 1.35, -1, 1, room, 0)
    _lib.check(l.gsv_t2s_decode(eng._h, C.byref(sp), None, 0, out.data_ptr(), ol.data_ptr(), C.byref(steps), s))
    assert steps.value == room


# ---------------------------------------------------------------------------------------------------------------------
# more than 32 rows: up to four QUADS of rows per group, phase by phase one after the other inside the one launch
# (t2s_mega_kernel<true>; VERDICT r2 next-round 5: BASELINE configs[2]'s 128 utterances per GPU in one engine call)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def v2_engine_128():
    cfg, sd = _v2()
    return _engine(cfg, sd, max_batch=128, max_seq=320)


@pytest.mark.parametrize("B", [33, 40, 64, 100, 128])
def test_multi_quad_one_pass_logits_match_launch_path(v2_engine_128, B):
    eng = v2_engine_128
    xs, berts, prompts = _batch(B)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=1, repetition_penalty=1.35)
    eng.set_mega(True)
    eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 1, "the persistent engine must take batches up to 128"
    la = eng.debug_logits(B).cpu().numpy()
    eng.set_mega(False)
    eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    lb = eng.debug_logits(B).cpu().numpy()
    eng.set_mega(True)
    err = np.abs(la - lb).max()
    print(f"[mega] B={B} (quads): one-pass logits vs launch path max-abs {err:.3e}")
    assert np.isfinite(la).all() and err < 3e-2


def test_multi_quad_ids_equal_the_single_quad_engine_row_by_row(v2_engine_128, v2_engine):
    """128 utterances in ONE engine call give, row by row, the ids the same utterances give in four calls of 32 (same
    arithmetic per row whatever its quad: MFMA columns, attention, LayerNorm and sampling are per row), 60 greedy tokens."""
    xs, berts, prompts = _batch(128)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=60, repetition_penalty=1.35)
    big = v2_engine_128
    big.set_mega(True)
    ya, ia = big.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    mode, ms, steps = big.decode_info()
    assert mode == 1 and ia == [60] * 128
    print(f"[mega] B=128 in one call: {ms:.2f} ms for {steps} steps = {ms / steps * 1e3:.1f} us per step")
    ya2, _ = big.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert [t.tolist() for t in ya] == [t.tolist() for t in ya2], "deterministic"
    small = v2_engine
    small.set_mega(True)
    tot = 0.0
    for lo in range(0, 128, 32):
        yb, ib = small.infer_panel_batch_infer(xs[lo:lo + 32], None, prompts[lo:lo + 32], berts[lo:lo + 32], **kw)
        assert small.decode_info()[0] == 1
        tot += small.decode_info()[1]
        for j in range(32):
            assert ya[lo + j].tolist() == yb[j].tolist(), f"row {lo + j} differs between the one-call and the four-call run"
    print(f"[mega] the same 128 rows as four calls of 32: {tot:.2f} ms of engine time ({tot / ms:.2f} x the one call)")


def test_multi_quad_ragged_eos_and_teacher_forced_logits():
    """B = 77 (ten groups' worth of ragged quads: groups hold 10 or 9 rows, the last quad 1 or 2), ragged text lengths, rows that
    finish by EOS at different steps while their quad mates go on; and teacher-forced per-step logits vs the launch path."""
    from gsv import synthetic as S
    B = 77
    cfg, sd = _v2(seed=3, suppress_eos=True)
    utt = S.make_utterances(B)
    g = torch.Generator().manual_seed(B)
    xs = [torch.tensor(it["all_phones"][:20 + int(torch.randint(0, 60, (1,), generator=g))], device=DEV) for it in utt["items"]]
    berts = [None] * B
    prompts = utt["prompt_semantic"].unsqueeze(0).expand(B, -1).contiguous().to(DEV)
    P = prompts.shape[1]
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=40, repetition_penalty=1.35)
    probe = _engine(cfg, sd, max_batch=128, max_seq=400)
    probe.set_mega(False)
    y0, _ = probe.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    gen = [y[P:].tolist() for y in y0]
    # the trigger token: first occurrences spread over many different steps, none in the first two
    best, tstar = -1, None
    for tok in sorted(set(gen[0][2:])):
        firsts = [gg.index(tok) if tok in gg else 40 for gg in gen]
        if min(firsts) >= 2 and len(set(firsts)) > best:
            best, tstar = len(set(firsts)), tok
    sd = dict(sd)
    w = sd["ar_predict_layer.weight"].clone()
    w[1024] = 1.05 * w[tstar]                       # a clear margin: EOS leads the trigger token by 5 % of the logit
    sd["ar_predict_layer.weight"] = w
    eng = _engine(cfg, sd, max_batch=128, max_seq=400)
    eng.set_mega(False)
    yb, ib = eng.infer_panel_batch_infer(xs, None, prompts, berts, dump_logits=True, **kw)
    Lb = eng.last_logits_dump.cpu().numpy()
    eng.set_mega(True)
    ya, ia = eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    assert eng.decode_info()[0] == 1
    same = sum(a.tolist() == b.tolist() and na == nb for a, b, na, nb in zip(ya, yb, ia, ib))
    print(f"[mega] B=77 (quads) ragged EOS: {same}/77 rows identical to the launch path; {len(set(ia))} different finish steps")
    assert len(set(ia)) > 3 and min(ia) < 40
    assert same >= 70
    for a, na in zip(ya, ia):
        assert a.shape[0] == P + na
    # teacher forcing on the launch path's tokens: per-step logits of rows still running
    tok = torch.zeros(B, 41, dtype=torch.int32)
    for r, (y, n) in enumerate(zip(yb, ib)):
        tok[r, :n] = y[P:].to(torch.int32).cpu()
        if n < 41:
            tok[r, n:] = 1024                       # the finishing token: the forced run ends where the launch path ended
    eng.infer_panel_batch_infer(xs, None, prompts, berts, force_tokens=tok, dump_logits=True, **kw)
    assert eng.decode_info()[0] == 1
    La = eng.last_logits_dump.cpu().numpy()
    worst = 0.0
    for r, n in enumerate(ib):
        worst = max(worst, float(np.abs(La[:n + 1, r] - Lb[:n + 1, r]).max()))
    print(f"[mega] B=77 (quads) teacher-forced logits vs launch path over every executed step: max-abs {worst:.3e}")
    assert worst <= 4e-2
