import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gpt-sovits_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from gsv import synthetic as S
from gsv.AR.models.t2s_model import Text2SemanticDecoder
DEV='cuda:0'
cfg = S.T2S_V2_CONFIG
sd = S.make_t2s_state_dict(cfg, seed=0, suppress_eos=True)
eng = Text2SemanticDecoder(cfg, device=DEV, dtype=torch.float16, max_batch=128, max_seq=320); eng.load_state_dict(sd)
for B in (33, 40):
    utt = S.make_utterances(B)
    xs = [torch.tensor(it["all_phones"], device=DEV) for it in utt["items"]]
    berts = [None]*B
    prompts = utt["prompt_semantic"].unsqueeze(0).expand(B, -1).contiguous().to(DEV)
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, early_stop_num=1, repetition_penalty=1.35)
    eng.set_mega(True); eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw); la = eng.debug_logits(B).cpu().numpy(); print(eng.decode_info(), eng.engine_stats())
    eng.set_mega(False); eng.infer_panel_batch_infer(xs, None, prompts, berts, **kw); lb = eng.debug_logits(B).cpu().numpy()
    e = np.abs(la-lb).max(1)
    print(B, " ".join(f"{i}:{v:.3f}" for i, v in enumerate(e)))
