"""Which component shows the intermittent 20-35 ms gaps?  Times 40 back-to-back calls of (a) the AR engine alone,
(b) the SoVITS decoder alone, (c) a torch-only D2H copy loop of the same 8 MB, each followed by a stream sync."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpt-sovits_amd"))
import torch  # noqa: E402

from gsv import synthetic as S  # noqa: E402
from gsv.AR.models.t2s_model import Text2SemanticDecoder  # noqa: E402
from gsv.module.models import SynthesizerTrn  # noqa: E402


def loop(fn, n=40):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    return ts


def main():
    B, TOK = 32, 100
    cfg = dict(S.T2S_V2_CONFIG)
    sd = S.make_t2s_state_dict(cfg, seed=0, suppress_eos=True)
    utt = S.make_utterances(B, seed=0)
    xs = [torch.tensor(utt["prompt_phones"] + it["phones"]) for it in utt["items"]]
    prompts = utt["prompt_semantic"].view(1, -1).expand(B, -1)
    eng = Text2SemanticDecoder(cfg, device="cuda:0", dtype=torch.float16, max_batch=B, max_seq=512)
    eng.load_state_dict(sd)
    ar = lambda: eng._run(xs, prompts, [None] * B, 1, 1.0, TOK - 1, 1.0, 1.35, eos_mask_steps=1, seed=0)
    ar()
    print(json.dumps({"ar_ms": loop(ar)}), flush=True)
    vcfg = S.VITS_V2_CONFIG
    vsd = S.make_vits_state_dict(vcfg, seed=0)
    d = vcfg["data"]
    v = SynthesizerTrn(d["filter_length"] // 2 + 1, vcfg["train"]["segment_size"] // d["hop_length"], n_speakers=d["n_speakers"],
                       version="v2", device="cuda:0", dtype=torch.float16, n_symbols=vcfg["n_symbols"], **vcfg["model"])
    v.load_state_dict(vsd)
    codes = torch.from_numpy(S.hash_ints("c", B * TOK, 1024, 0)).view(1, 1, -1).cuda()
    text = torch.from_numpy(S.hash_ints("t", B * 40, vcfg["n_symbols"], 0)).view(1, -1).cuda()
    refer = [S.make_refer_spec().cuda()]
    dec = lambda: v.decode(codes, text, refer, seed=1)
    dec()
    print(json.dumps({"sovits_ms": loop(dec)}), flush=True)
    big = torch.zeros(4 * 1024 * 1024, dtype=torch.int16, device="cuda:0")

    def d2h():
        big.add_(1)
        return big.cpu()
    d2h()
    print(json.dumps({"d2h_ms": loop(d2h)}), flush=True)


if __name__ == "__main__":
    main()
