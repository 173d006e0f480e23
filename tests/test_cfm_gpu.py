"""GPU: the HIP flow-matching mel decoder (H14: CFM.inference over the DiT, through the C ABI) against the
reference classes' output (golden, oracle/gen_golden_vits.py::gen_cfm) and the oracle.  The rotary embedding on
both sides of the golden comparison is oracle/rope.py's restatement of x_transformers (parity unpinned there)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfm(cfg, sd, dtype):
    from gsv.f5_tts.model.backbones.dit import DiT
    from gsv.module.models import CFM
    dit = DiT(dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"], ff_mult=cfg["ff_mult"],
              mel_dim=cfg["mel_dim"], text_dim=cfg["text_dim"], conv_layers=cfg["conv_layers"], device=DEV, dtype=dtype)
    dit.load_state_dict({"cfm.estimator." + k: v for k, v in sd.items()})
    return CFM(cfg["mel_dim"], dit)


@pytest.mark.parametrize("name", list(cases.CFM_CASES))
def test_cfm_fp32_matches_reference(name):
    """fp32 engine vs the reference's CFM.inference output with the same injected noise: max-abs <= 2e-3 on mels of
    rms ~1.2 (fp32 MFMA products summed in a different order through depth x steps residual blocks)."""
    case = cases.CFM_CASES[name]
    cfg, sd, mu, prompt, noise = cases.cfm_case_inputs(case)
    g = load_golden(name)["mel"]
    out = _cfm(cfg, sd, torch.float32).inference(mu.to(DEV), None, prompt.to(DEV), case["steps"], noise=noise).cpu().numpy()
    assert out.shape == g.shape
    assert np.all(out[:, :, :case["Tp"]] == 0)                      # prompt frames are zeroed (models.py:1084)
    assert np.abs(out - g).max() <= 2e-3


@pytest.mark.parametrize("name", ["cfm_small", "cfm_v3dims"])
def test_cfm_fp16_within_tolerance(name):
    """fp16 engine (production dtype) vs the fp32 reference: relative RMS error <= 3 %, max-abs <= 0.15."""
    case = cases.CFM_CASES[name]
    cfg, sd, mu, prompt, noise = cases.cfm_case_inputs(case)
    g = load_golden(name)["mel"]
    out = _cfm(cfg, sd, torch.float16).inference(mu.to(DEV), None, prompt.to(DEV), case["steps"], noise=noise).float().cpu().numpy()
    err = out - g
    assert np.sqrt((err ** 2).mean()) <= 0.03 * np.sqrt((g ** 2).mean())
    assert np.abs(err).max() <= 0.15


def test_cfm_device_noise_and_errors():
    """noise=None draws N(0,1) on the device from the seed (deterministic per seed, different across seeds); bad
    shapes and the unsupported CFG branch raise."""
    case = cases.CFM_CASES["cfm_small"]
    cfg, sd, mu, prompt, noise = cases.cfm_case_inputs(case)
    cfm = _cfm(cfg, sd, torch.float32)
    a = cfm.inference(mu.to(DEV), None, prompt.to(DEV), 2, seed=5)
    b = cfm.inference(mu.to(DEV), None, prompt.to(DEV), 2, seed=5)
    c = cfm.inference(mu.to(DEV), None, prompt.to(DEV), 2, seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()
    with pytest.raises(ValueError):
        cfm.inference(mu[:, :, :8].to(DEV), None, prompt.to(DEV), 2)
    with pytest.raises(ValueError):
        cfm.inference(mu[:, :4].to(DEV), None, prompt.to(DEV), 2)          # prompt longer than the sequence
    with pytest.raises(NotImplementedError):
        cfm.inference(mu.to(DEV), None, prompt.to(DEV), 2, inference_cfg_rate=0.5)
