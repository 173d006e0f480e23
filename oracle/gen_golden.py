"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REFERENCE
(/root/reference, imported through oracle/ref_import.py) on this repo's synthetic
checkpoints and inputs, and checks the oracle restatement against it on the way.

Run in the build container only:  python oracle/gen_golden.py [t2s] [vits] [aa]
The fixtures hold inputs (as generator seeds/specs or small arrays) and the
reference's outputs -- data only, no reference source.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from gsv import synthetic as S  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.t2s_oracle import T2SOracle  # noqa: E402


from oracle.cases import T2S_CASES, t2s_case_inputs  # noqa: E402


def run_reference_t2s(cfg, sd, xs, berts, prompts, noise, case, naive=False):
    ref_import.setup()
    import AR.models.utils as ru
    import AR.models.t2s_model as rt
    model = ref_import.t2s_decoder_cls()(cfg)
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    step = {"i": 0}
    logits_trace = []
    orig_mn = ru.multinomial_sample_one_no_sync
    orig_sample = rt.sample

    def patched_mn(probs):
        q = noise[step["i"]][:, : probs.shape[1]].expand_as(probs)
        return torch.argmax(probs / q, dim=-1, keepdim=True).to(dtype=torch.int)

    def traced_sample(logits, previous_tokens=None, **kw):
        logits_trace.append(logits.detach().clone())
        out = orig_sample(logits, previous_tokens, **kw)
        step["i"] += 1
        return out

    if noise is not None:
        ru.multinomial_sample_one_no_sync = patched_mn
    rt.sample = traced_sample
    try:
        with torch.no_grad():
            kw = dict(top_k=case["top_k"], top_p=case["top_p"], temperature=case["temperature"],
                      early_stop_num=case["early_stop"], repetition_penalty=case["rep"])
            if naive:
                y, idx = model.infer_panel_naive(xs[0].unsqueeze(0), torch.LongTensor([xs[0].shape[0]]),
                                                 prompts[:1], berts[0].unsqueeze(0), **kw)
                ys, idxs = [y[0]], [int(idx)]
            else:
                ys, idxs = model.infer_panel_batch_infer(xs, torch.LongTensor([t.shape[0] for t in xs]), prompts,
                                                         berts, **kw)
    finally:
        ru.multinomial_sample_one_no_sync = orig_mn
        rt.sample = orig_sample
    return [t.clone() for t in ys], [int(i) for i in idxs], logits_trace


def gen_t2s_ref_free():
    """prompt-free naive decode (reference t2s_model.py:849-856, 916-917): reference vs oracle, greedy and sampled"""
    for name in ("t2s_small_greedy", "t2s_small_topk"):
        case = T2S_CASES[name]
        cfg, sd, xs, berts, prompts, noise = t2s_case_inputs(case)
        ref_import.setup()
        import AR.models.utils as ru
        model = ref_import.t2s_decoder_cls()(cfg)
        model.load_state_dict(sd, strict=True)
        model.eval()
        step = {"i": 0}
        orig_mn = ru.multinomial_sample_one_no_sync

        def patched_mn(probs):
            q = noise[step["i"]][:, : probs.shape[1]].expand_as(probs)
            step["i"] += 1
            return torch.argmax(probs / q, dim=-1, keepdim=True).to(dtype=torch.int)

        if noise is not None:
            ru.multinomial_sample_one_no_sync = patched_mn
        kw = dict(top_k=case["top_k"], top_p=case["top_p"], temperature=case["temperature"],
                  early_stop_num=case["early_stop"], repetition_penalty=case["rep"])
        try:
            with torch.no_grad():
                y, idx = model.infer_panel_naive(xs[0].unsqueeze(0), torch.LongTensor([xs[0].shape[0]]), None,
                                                 berts[0].unsqueeze(0), **kw)
        finally:
            ru.multinomial_sample_one_no_sync = orig_mn
        nz = None if noise is None else noise[:, :1]
        oy, oidx = T2SOracle(sd, cfg).infer_panel_naive(xs[0].unsqueeze(0), None, None, berts[0].unsqueeze(0), noise=nz, **kw)
        ok = int(idx) == oidx == 0 and torch.equal(y[0].long(), oy[0].long())
        print(f"[gen_golden] {name}_ref_free: tokens {y.shape[1]} idx={int(idx)} oracle_match={ok}")
        assert ok
        np.savez_compressed(os.path.join(GOLD, name + "_ref_free.npz"), y=y[0].numpy().astype(np.int64))


def gen_t2s():
    for name, case in T2S_CASES.items():
        cfg, sd, xs, berts, prompts, noise = t2s_case_inputs(case)
        for naive in (False, True):
            ys, idxs, ltrace = run_reference_t2s(cfg, sd, xs, berts, prompts, noise, case, naive=naive)
            orc = T2SOracle(sd, cfg)
            B = 1 if naive else len(xs)
            nz = None if noise is None else noise.expand(-1, B, -1)
            trace = {}
            kw = dict(top_k=case["top_k"], top_p=case["top_p"], temperature=case["temperature"],
                      early_stop_num=case["early_stop"], repetition_penalty=case["rep"], noise=nz)
            if naive:
                y, idx = orc.infer_panel_naive(xs[0].unsqueeze(0), None, prompts[:1], berts[0].unsqueeze(0),
                                               trace=trace, **kw)
                oys, oidx = [y[0]], [idx]
            else:
                oys, oidx = orc.infer_panel_batch_infer(xs, None, prompts, berts, trace=trace, **kw)
            ok = oidx == idxs and all(torch.equal(a, b) for a, b in zip(oys, ys))
            # top-2 margin of step-0 logits (how far greedy ids are from a tie)
            l0 = ltrace[0]
            top2 = torch.topk(l0, 2, dim=-1).values
            dl = (trace["logits"][0][: l0.shape[0]] - l0).abs().max().item()
            tag = name + ("_naive" if naive else "")
            print(f"[gen_golden] {tag}: idx={idxs} oracle_match={ok} step0 |dlogits|max={dl:.2e} "
                  f"min top2 margin={float((top2[:, 0] - top2[:, 1]).min()):.3f}")
            assert ok, f"oracle restatement disagrees with the reference on {tag}"
            # per-step reference top-2 values/ids of the first live row (small)
            margins = np.array([float((torch.topk(l, 2, -1).values[:, 0] - torch.topk(l, 2, -1).values[:, 1]).min())
                                for l in ltrace], dtype=np.float32)
            out = dict(
                y_flat=np.concatenate([t.numpy() for t in ys]).astype(np.int64),
                y_lens=np.array([t.shape[0] for t in ys], dtype=np.int64),
                idx=np.array(idxs, dtype=np.int64),
                step0_logits=ltrace[0].numpy().astype(np.float32),
                min_top2_margin=margins,
            )
            np.savez_compressed(os.path.join(GOLD, tag + ".npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["t2s", "vits", "aa", "voc", "cfm", "encp"]
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "t2s" in what:
        gen_t2s()
    if "vits" in what:
        from oracle.gen_golden_vits import gen_vits
        gen_vits()
    if "aa" in what:
        from oracle.gen_golden_vits import gen_aa
        gen_aa()
    if "voc" in what:
        from oracle.gen_golden_vits import gen_voc
        gen_voc()
    if "reffree" in what:
        gen_t2s_ref_free()
    if "encp" in what:
        from oracle.gen_golden_vits import gen_encp
        gen_encp()
    if "cfm" in what:
        from oracle.gen_golden_vits import gen_cfm
        gen_cfm()
