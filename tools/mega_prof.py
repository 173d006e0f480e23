#!/usr/bin/env python3
"""In-kernel time line of one layer of the persistent AR decode engine (measurement tool).

Runs the BASELINE configs[1] AR workload (B = 32, 100 tokens) with GSV_MEGA_PROF set, so that every wave of every workgroup
stamps the 100 MHz real-time counter at the phase boundaries of one (step, layer), and prints the median / max offset of each
stamp from the layer's start for the comm role (wave 0) and the compute role (wave 4).

    python tools/mega_prof.py [out.txt]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpt-sovits_amd")):
    sys.path.insert(0, p)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mega_prof_raw.txt")
if os.environ.get("PROF_OFF") != "1":            # PROF_OFF=1: timing only (the stamps themselves cost ~15 us per step)
    os.environ["GSV_MEGA_PROF"] = out

import torch  # noqa: E402
from gsv import synthetic as S  # noqa: E402
if os.environ.get("GSV_LIB_PATH"):                 # A/B against another build of the library
    from gsv import _lib
    _lib.LIB_PATH = os.environ["GSV_LIB_PATH"]
from gsv.AR.models.t2s_model import Text2SemanticDecoder  # noqa: E402

B = int(os.environ.get("PROF_B", "32"))
cfg = S.T2S_V2_CONFIG
eng = Text2SemanticDecoder(cfg, device="cuda:0", dtype=torch.float16, max_batch=max(32, B), max_seq=320)
eng.load_state_dict(S.make_t2s_state_dict(cfg, seed=0, suppress_eos=True))
utt = S.make_utterances(B)
xs = [torch.tensor(it["all_phones"], device="cuda:0") for it in utt["items"]]
berts = [it["bert"].to("cuda:0") for it in utt["items"]]
prompts = utt["prompt_semantic"].unsqueeze(0).expand(B, -1).contiguous().to("cuda:0")
kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=100, repetition_penalty=1.35)
for _ in range(3):
    eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
mode, ms, steps = eng.decode_info()
print(f"mode {mode}: {ms:.2f} ms for {steps} steps = {ms / steps * 1e3:.1f} us/step = {ms / steps * 1e3 / 24.5:.2f} us/layer")
if os.environ.get("PROF_OFF") == "1":
    sys.exit(0)
rows = [l.split() for l in open(out) if not l.startswith("#")]
COMM = ["start", "hopA+LN done", "B1", "B2", "B3", "B4", "hopB done", "B1", "B2", "hopC+LN done", "B1", "B2", "hopD done", "B1", "B2"]
COMP = ["start", "wB issued", "B1", "P1 gemm", "B2", "reduce/append", "kv vmcnt0", "B3", "attention", "B4", "(unused)", "publish B",
        "B1", "P2 gemm", "B2", "pub C, wD issued", "B1", "P3 gemm", "B2", "pub D, wA0 issued", "B1", "P4 gemm", "B2", "pub A"]
for role, wave, names in (("comm (wave 0)", 0, COMM), ("compute (wave 4)", 4, COMP)):
    sel = [r for r in rows if int(r[1]) == wave]
    print(f"--- {role}: {len(sel)} workgroups; offset from the layer's start in us (median, max), delta to previous stamp")
    prev = 0.0
    for i, n in enumerate(names):
        if i == 0:
            continue
        vals = [int(r[2 + i]) / 100.0 for r in sel if int(r[2 + i]) >= 0]
        if not vals:
            continue
        med = statistics.median(vals)
        print(f"  {i:2d} {n:22s} {med:7.2f} {max(vals):7.2f}   +{med - prev:5.2f}")
        prev = med
