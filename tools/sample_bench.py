#!/usr/bin/env python3
"""Cost of the sampling kernel (`gsv_op_sample` = sample_core of csrc/t2s_sample.h, the code the decode step and the persistent
engine both run) at the reference's parameter sets: greedy, the CLI default (top_k 5), the web UI defaults (top_k 15 / 20),
and top-p < 1, where the kept set is extracted by repeated wave arg-max rounds (measurement tool, not product code).

    python tools/sample_bench.py > profiles/r02_sampling.json
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpt-sovits_amd"))
import torch  # noqa: E402

from gsv import _lib  # noqa: E402

B, V, PREV = 32, 1025, 230
_lib.init(0)
g = torch.Generator().manual_seed(0)
logits = (torch.randn(B, V, generator=g) * 3.0).cuda()
prev = torch.randint(0, 1024, (B, PREV), generator=g, dtype=torch.int32).cuda()
smp = torch.zeros(B, dtype=torch.int32, device="cuda")
amx = torch.zeros(B, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for top_k, top_p, temp in ((1, 1.0, 1.0), (5, 1.0, 1.0), (15, 1.0, 1.0), (20, 1.0, 1.0), (15, 0.9, 1.0), (20, 0.6, 1.0), (0, 0.9, 1.0),
                           (0, 0.6, 1.3), (100, 1.0, 1.0)):
    sp = _lib.SamplingParams(top_k, top_p, temp, 1.35, -1, 0, 1500, 0)
    call = lambda step: _lib.check(_lib.lib().gsv_op_sample(logits.data_ptr(), B, V, V, prev.data_ptr(), PREV, C.byref(sp), None, step,
                                                            smp.data_ptr(), amx.data_ptr(), st), "gsv_op_sample")
    for i in range(5):
        call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for i in range(n):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"B": B, "V": V, "top_k": top_k, "top_p": top_p, "temperature": temp, "repetition_penalty": 1.35,
                      "us_per_launch_back_to_back": round(e0.elapsed_time(e1) * 1e3 / n, 2)}))
