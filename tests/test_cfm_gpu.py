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


def _v3(cfg, sd, version, dtype):
    from gsv.module.models import SynthesizerTrnV3
    from gsv import synthetic as S
    d = cfg["data"]
    dit = S.small_dit_config()
    dit["text_dim"] = 512
    m = SynthesizerTrnV3(d["filter_length"] // 2 + 1, cfg["train"]["segment_size"] // d["hop_length"],
                         n_speakers=d["n_speakers"], version=version, device=DEV, dtype=dtype, n_symbols=cfg["n_symbols"],
                         dit_kwargs={k: v for k, v in dit.items() if k != "mel_dim"}, **cfg["model"])
    full = dict(sd)
    full.update({"cfm.estimator." + k: v for k, v in S.make_dit_state_dict(dit, seed=0).items()})
    m.load_state_dict(full)
    return m


@pytest.mark.parametrize("name", list(cases.ENCP_CASES))
def test_decode_encp_matches_reference(name):
    """SynthesizerTrnV3.decode_encp (enc_p -> bridge -> nearest x1.875 | x2 -> wns1 with its tail mask), fp32 engine vs
    the reference class (golden): max-abs <= 5e-4 on features of rms ~1.1; fp16 engine: relative RMS <= 2 %."""
    case = cases.ENCP_CASES[name]
    cfg, sd, codes, text, refer = cases.encp_case_inputs(case)
    g = load_golden(name)["fea"]
    m = _v3(cfg, sd, case["version"], torch.float32)
    fea, ge = m.decode_encp(codes.to(DEV), text.to(DEV), refer.to(DEV), speed=case["speed"])
    assert tuple(fea.shape) == g.shape
    assert np.abs(fea.cpu().numpy() - g).max() <= 5e-4
    fea2, ge2 = m.decode_encp(codes.to(DEV), text.to(DEV), refer.to(DEV), ge=ge, speed=case["speed"])     # cached ge
    assert torch.equal(fea, fea2) and ge2 == ge
    m16 = _v3(cfg, sd, case["version"], torch.float16)
    f16 = m16.decode_encp(codes.to(DEV), text.to(DEV), refer.to(DEV), speed=case["speed"])[0].float().cpu().numpy()
    assert np.sqrt(((f16 - g) ** 2).mean()) <= 0.02 * np.sqrt((g ** 2).mean())
    with pytest.raises(NotImplementedError):
        m.decode(codes.to(DEV), text.to(DEV), refer.to(DEV))


@pytest.mark.parametrize("B,T,Tp", [(1, 1, 0), (1, 5, 5), (2, 37, 0), (3, 20, 7)])
def test_cfm_edge_shapes_match_oracle(B, T, Tp):
    """single frame, prompt covering the whole sequence (nothing generated: all zeros), no prompt at all, and a batch
    whose ONE prompt is broadcast over the rows (as using_vocoder_synthesis_batched_infer passes it): fp32 engine vs the
    oracle, max-abs <= 2e-3."""
    from gsv import synthetic as S
    from oracle import cfm_oracle
    cfg = S.small_dit_config()
    sd = S.make_dit_state_dict(cfg, seed=5)
    mu = S.hash_symmetric("edge_mu", (B, T, cfg["text_dim"]), 1.0, B * 100 + T)
    prompt = S.hash_symmetric("edge_prompt", (1, cfg["mel_dim"], Tp), 1.0, T)
    noise = S.hash_normal("edge_noise", (B, cfg["mel_dim"], T), T)
    out = _cfm(cfg, sd, torch.float32).inference(mu.to(DEV), None, prompt.to(DEV), 3, noise=noise).cpu()
    ref = cfm_oracle.cfm_inference(sd, cfg, mu, prompt.expand(B, -1, -1), 3, noise.clone())
    assert out.shape == ref.shape == (B, cfg["mel_dim"], T)
    assert (out - ref).abs().max() <= 2e-3
    if Tp == T:
        assert float(out.abs().max()) == 0.0


def test_decode_encp_single_token():
    """one semantic token, one phoneme (2 enc_p frames -> 3 feature frames for v3): fp32 engine vs the oracle <= 5e-4."""
    from oracle.vits_oracle import VitsOracle
    case = dict(cases.ENCP_CASES["encp_small_v3"], T=1, L=1)
    cfg, sd, codes, text, refer = cases.encp_case_inputs(case)
    m = _v3(cfg, sd, "v3", torch.float32)
    fea, _ = m.decode_encp(codes.to(DEV), text.to(DEV), refer.to(DEV))
    ref, _ = VitsOracle(sd, cfg).decode_encp(codes, text, refer, speed=1, version="v3")
    assert tuple(fea.shape) == tuple(ref.shape) == (1, 512, 3)
    assert (fea.cpu() - ref).abs().max() <= 5e-4


def test_cfm_full_depth_at_chunk_length_vs_oracle():
    """BASELINE configs[3] shape: the full DiT (1024 x 22 blocks, 16 heads x 64) over one 934-frame chunk with a 468-frame
    prompt, 2 Euler steps, fp32 engine vs the CPU oracle (0.9 TFLOP on the host).  Depth and length are where a fused
    attention / split-K GEMM indexing error would show; max-abs <= 5e-4 on mels of rms ~1 (measured 7.6e-6)."""
    from gsv import synthetic as S
    from oracle import cfm_oracle
    cfg = dict(S.DIT_V3_CONFIG)
    sd = S.make_dit_state_dict(cfg, seed=9)
    B, T, Tp = 1, 934, 468
    mu = S.hash_symmetric("full_mu", (B, T, cfg["text_dim"]), 1.0, 1)
    prompt = S.hash_symmetric("full_prompt", (1, cfg["mel_dim"], Tp), 1.0, 2)
    noise = S.hash_normal("full_noise", (B, cfg["mel_dim"], T), 3)
    torch.set_num_threads(8)
    ref = cfm_oracle.cfm_inference(sd, cfg, mu, prompt, 2, noise.clone())
    out = _cfm(cfg, sd, torch.float32).inference(mu.to(DEV), None, prompt.to(DEV), 2, noise=noise).cpu()
    err = (out - ref).abs().max().item()
    print(f"[parity] depth-22 DiT, T=934, 2 steps: max-abs error {err:.2e} (mel rms {ref.pow(2).mean().sqrt():.3f})")
    assert out.shape == ref.shape == (1, 100, 934)
    assert err <= 5e-4
