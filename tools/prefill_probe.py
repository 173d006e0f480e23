#!/usr/bin/env python3
"""Wall time of the AR prefill (+ step 0 + one engine step) at the BASELINE configs[1] shape: B = 32 rows of 80 phonemes + 100
prompt tokens (measurement tool; `rocprofv3 --kernel-trace --stats -- python3 tools/prefill_probe.py` gives the kernel split).

    python tools/prefill_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpt-sovits_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from gsv import synthetic as S  # noqa: E402
from gsv.AR.models.t2s_model import Text2SemanticDecoder  # noqa: E402

B = int(os.environ.get("PROF_B", "32"))
cfg = S.T2S_V2_CONFIG
eng = Text2SemanticDecoder(cfg, device="cuda:0", dtype=torch.float16, max_batch=32, max_seq=320)
eng.load_state_dict(S.make_t2s_state_dict(cfg, seed=0, suppress_eos=True))
utt = S.make_utterances(B)
xs = [torch.tensor(it["all_phones"]) for it in utt["items"]]
berts = [it["bert"] for it in utt["items"]]
prompts = utt["prompt_semantic"].unsqueeze(0).expand(B, -1).contiguous()
kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=1, repetition_penalty=1.35)
ts = []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print("prefill + step 0 + 1 engine step, B = %d: median %.3f ms, min %.3f ms" % (B, 1e3 * ts[len(ts) // 2], 1e3 * ts[0]))
