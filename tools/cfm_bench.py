"""Times the flow-matching mel decoder (H14) at the reference's v3 shape: DiT dim 1024 x 22 blocks, one 934-frame chunk
with a 468-frame prompt (TTS.py:617-621), fp16.  Prints one JSON line: ms per Euler step and achieved TFLOP/s
(algorithmic flops: Linear layers + attention products + position conv)."""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpt-sovits_amd"))
import torch  # noqa: E402

from gsv import synthetic as S  # noqa: E402


def flops_per_step(cfg, T):
    D, inner, FF = cfg["dim"], cfg["heads"] * cfg["dim_head"], cfg["dim"] * cfg["ff_mult"]
    per_block = 2 * T * (D * 3 * inner + inner * D + 2 * D * FF) + 4 * T * T * inner
    pos = 2 * 2 * T * D * (D // 16) * 31
    inp = 2 * T * (2 * cfg["mel_dim"] + cfg["text_dim"]) * D
    return cfg["depth"] * per_block + pos + inp + 2 * T * D * cfg["mel_dim"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--frames", type=int, default=934)
    ap.add_argument("--prompt", type=int, default=468)
    ap.add_argument("--depth", type=int, default=22)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="chunks per call (the batched caller passes 4-8)")
    a = ap.parse_args()
    from gsv.f5_tts.model.backbones.dit import DiT
    from gsv.module.models import CFM
    cfg = dict(S.DIT_V3_CONFIG)
    cfg["depth"] = a.depth
    dt = torch.float32 if a.fp32 else torch.float16
    dit = DiT(dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"], ff_mult=cfg["ff_mult"],
              mel_dim=cfg["mel_dim"], text_dim=cfg["text_dim"], conv_layers=cfg["conv_layers"], device="cuda:0", dtype=dt)
    dit.load_state_dict(S.make_dit_state_dict(cfg, seed=1))
    cfm = CFM(100, dit)
    mu = S.hash_symmetric("bench_mu", (a.batch, a.frames, cfg["text_dim"]), 1.0, 1).cuda()
    prompt = S.hash_symmetric("bench_prompt", (1, 100, a.prompt), 1.0, 1).cuda()
    cfm.inference(mu, None, prompt, 2, seed=1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out = cfm.inference(mu, None, prompt, a.steps, seed=1)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    fl = flops_per_step(cfg, a.frames) * a.batch
    print(json.dumps({"what": "cfm_inference", "dtype": "f32" if a.fp32 else "f16", "frames": a.frames, "prompt": a.prompt,
                      "depth": a.depth, "batch": a.batch, "steps": a.steps, "ms_total": best * 1e3, "ms_per_step": best * 1e3 / a.steps,
                      "gflop_per_step": fl / 1e9, "tflops": fl * a.steps / best / 1e12,
                      "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
