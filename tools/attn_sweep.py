"""Decode-attention kernel (the HBM-bound kernel of the AR step) timed at several (B, S) points:
shows how its achieved bandwidth depends on the bytes one launch moves.  Prints a JSON list."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
import torch  # noqa: E402
from gsv import _lib, synthetic as S  # noqa: E402
from gsv.AR.models.t2s_model import Text2SemanticDecoder  # noqa: E402

eng = Text2SemanticDecoder(S.T2S_V2_CONFIG, device="cuda:0", dtype=torch.float16, max_batch=128, max_seq=2064)
eng.load_state_dict(S.make_t2s_state_dict(S.T2S_V2_CONFIG, seed=0))
out = []
for B, Sx in [(1, 280), (8, 280), (32, 180), (32, 280), (32, 600), (32, 1500), (64, 280), (128, 280), (128, 600), (128, 2000)]:
    _lib.check(_lib.lib().gsv_t2s_debug_set_state(eng._h, B, Sx))
    ms, ab, total, step = eng.time_attention(iters=10)
    gbs = ab / (ms * 1e-3) / 1e9
    out.append({"B": B, "kv_len": Sx, "bytes_per_launch": ab, "avg_launch_us": round(ms * 1e3, 2),
                "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3),
                "step_layers_eager_ms": round(step, 3), "step_bytes": total,
                "step_frac_of_8TBps": round(total / (step * 1e-3) / 8e12, 3)})
    print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r01_attn_sweep.json"), "w"), indent=1)
