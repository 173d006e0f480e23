"""Probe: does splitting the AR batch over k engines on k streams (k host threads) shorten the decode?
The decode step is a latency-bound chain of 122 small launches; k independent chains can overlap on the chip."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpt-sovits_amd"))
import torch  # noqa: E402

from gsv import synthetic as S  # noqa: E402
from gsv.AR.models.t2s_model import Text2SemanticDecoder  # noqa: E402


def main():
    B, TOK = 32, 100
    cfg = dict(S.T2S_V2_CONFIG)
    sd = S.make_t2s_state_dict(cfg, seed=0, suppress_eos=True)
    utt = S.make_utterances(B, seed=0)
    xs = [torch.tensor(utt["prompt_phones"] + it["phones"]) for it in utt["items"]]
    prompts = utt["prompt_semantic"].view(1, -1).expand(B, -1)
    for k in (1, 2, 4):
        n = B // k
        engs = []
        for i in range(k):
            e = Text2SemanticDecoder(cfg, device="cuda:0", dtype=torch.float16, max_batch=n, max_seq=512)
            e.load_state_dict(sd)
            engs.append(e)
        res = [None] * k

        def work(i):
            res[i] = engs[i]._run(xs[i * n:(i + 1) * n], prompts[i * n:(i + 1) * n], [None] * n, 1, 1.0, TOK - 1, 1.0, 1.35,
                                  eos_mask_steps=1, seed=0)

        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        toks = sum(sum(r[1]) for r in res)
        print(json.dumps({"engines": k, "rows_each": n, "ms": round(best * 1e3, 2), "tokens": toks}), flush=True)
        del engs


if __name__ == "__main__":
    main()
