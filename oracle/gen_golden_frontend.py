"""TEST INFRASTRUCTURE ONLY -- fixtures for the reference-audio front-end, build container only.

  tests/golden/spec_*.npz     the reference's own `spectrogram_torch` (GPT_SoVITS/module/mel_processing.py:40-74, pure torch: it
                              imports here with oracle/ref_import's librosa stub) on seeded waveforms
  tests/golden/hubert_base.npz `transformers.HubertModel(HubertConfig())` -- the class the reference's CNHubert wraps
                              (feature_extractor/cnhubert.py:22-37) -- with gsv.synthetic.make_hubert_state_dict weights on a
                              seeded 1.3 s waveform: last_hidden_state (fp16) and a few intermediate checksums.
                              The container has transformers 5.x, the reference pins >= 4.43, <= 4.50 (requirements.txt:21):
                              same architecture and state-dict schema; recorded here because the oracle is that package.

    python oracle/gen_golden_frontend.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
from gsv import synthetic as S  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SPEC_CASES = {"spec_32k_half_s": (16000, 0), "spec_32k_ragged": (22001, 1), "spec_short": (1500, 2)}
HUBERT_N = 20800          # 1.3 s at 16 kHz -> 64 frames


def main():
    from transformers import HubertConfig, HubertModel
    torch.manual_seed(0)
    m = HubertModel(HubertConfig()).eval()
    sd = S.make_hubert_state_dict(seed=0)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    assert all("masked_spec_embed" in k for k in missing.missing_keys), missing.missing_keys      # training-only parameter
    wav = S.make_waveform(HUBERT_N, 7).unsqueeze(0)
    with torch.no_grad():
        out = m(wav, output_hidden_states=True)
        feats = m.feature_extractor(wav)                 # [1, 512, T]
    h = out["last_hidden_state"][0]
    np.savez_compressed(os.path.join(GOLD, "hubert_base.npz"), last_hidden_state=h.numpy().astype(np.float16),
                        conv_features_rms=np.float32(feats.pow(2).mean().sqrt().item()),
                        layer_rms=np.array([x.pow(2).mean().sqrt().item() for x in out["hidden_states"]], dtype=np.float32))
    print("hubert", tuple(h.shape), "rms", float(h.pow(2).mean().sqrt()))
    # the reference's module needs oracle/ref_import's third-party stubs, which would break the transformers import above
    ref_import.setup()
    from module.mel_processing import spectrogram_torch
    for name, (n, seed) in SPEC_CASES.items():
        y = S.make_waveform(n, seed, sr=32000).unsqueeze(0)
        spec = spectrogram_torch(y, 2048, 32000, 640, 2048, center=False)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), spec=spec.numpy().astype(np.float32))
        print(name, tuple(spec.shape), float(spec.max()))


if __name__ == "__main__":
    main()
