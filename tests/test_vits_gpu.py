"""GPU: the HIP SoVITS decoder (through the C ABI) against the reference's golden waveforms and
the oracle; BigVGAN anti-alias activation against its golden."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(cfg, sd, dtype):
    from gsv.module.models import SynthesizerTrn
    d = cfg["data"]
    mk = dict(cfg["model"])
    version = mk.pop("version", "v2")
    m = SynthesizerTrn(d["filter_length"] // 2 + 1, cfg["train"]["segment_size"] // d["hop_length"],
                       n_speakers=d["n_speakers"], version=version, device=DEV, dtype=dtype, n_symbols=cfg["n_symbols"],
                       **mk)
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("name", ["vits_small", "vits_small_2ref", "vits_v2", "vits_small_speed"])
def test_fp32_waveform_matches_reference(name):
    """fp32 engine vs the reference's waveform (golden), same injected noise: max-abs <= 1e-4
    (tolerance stated in BASELINE.md section 3); intermediates ge / m_p / z within 1e-4."""
    case = cases.VITS_CASES[name]
    cfg, sd, codes, text, refers, noise, ssl = cases.vits_case_inputs(case)
    g = load_golden(name)
    eng = _engine(cfg, sd, torch.float32)
    wav = eng.decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers], noise_scale=case["noise_scale"],
                     noise=noise, speed=case.get("speed", 1))
    assert tuple(wav.shape) == g["wav"].shape
    IC, F = cfg["model"]["inter_channels"], noise.shape[1]
    ge = eng.debug_tensor("ge", 512).cpu().numpy()
    assert np.abs(ge - g["ge"].reshape(-1)).max() < 1e-4
    m_p = eng.debug_tensor("m_p", IC * F).cpu().numpy().reshape(IC, F)
    assert np.abs(m_p - g["m_p"]).max() < 2e-4
    z = eng.debug_tensor("z", IC * F).cpu().numpy().reshape(IC, F)
    assert np.abs(z - g["z"]).max() < 5e-4
    assert np.abs(wav.float().cpu().numpy() - g["wav"]).max() <= 1e-4
    lat = eng.extract_latent(ssl.to(DEV))
    assert lat.cpu().numpy().tolist() == g["latent_codes"].tolist()      # integer codes: bit-exact


@pytest.mark.parametrize("name", ["vits_small", "vits_v2"])
def test_fp16_waveform_within_tolerance(name):
    """fp16 engine (production dtype) vs the fp32 reference waveform: max-abs <= 2e-2 and
    relative RMS error <= 3 % (fp16 storage of activations through ~100 conv layers)."""
    case = cases.VITS_CASES[name]
    cfg, sd, codes, text, refers, noise, ssl = cases.vits_case_inputs(case)
    g = load_golden(name)
    eng = _engine(cfg, sd, torch.float16)
    wav = eng.decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers], noise_scale=case["noise_scale"],
                     noise=noise).float().cpu().numpy()
    err = wav - g["wav"]
    assert np.abs(err).max() <= 2e-2
    assert np.sqrt((err ** 2).mean()) <= 0.03 * np.sqrt((g["wav"] ** 2).mean())


def test_decode_edge_cases_and_determinism():
    case = cases.VITS_CASES["vits_small"]
    cfg, sd, codes, text, refers, noise, ssl = cases.vits_case_inputs(case)
    eng = _engine(cfg, sd, torch.float32)
    # single token / single phoneme (shortest legal input)
    w1 = eng.decode(codes[:, :, :1].to(DEV), text[:, :1].to(DEV), refers[0].to(DEV), noise=noise[:, :2])
    assert w1.shape == (1, 1, 2 * 16) and torch.isfinite(w1).all()
    # empty inputs are rejected loudly
    with pytest.raises(ValueError):
        eng.decode(codes[:, :, :0].to(DEV), text.to(DEV), refers[0].to(DEV))
    # counter-RNG noise: same seed -> same waveform, different seed -> different
    a = eng.decode(codes.to(DEV), text.to(DEV), refers[0].to(DEV), seed=3)
    b = eng.decode(codes.to(DEV), text.to(DEV), refers[0].to(DEV), seed=3)
    c = eng.decode(codes.to(DEV), text.to(DEV), refers[0].to(DEV), seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # time-axis concatenation property used by TTS.run (TTS.py:1259-1282): output length is
    # tokens * 2 * prod(upsample_rates)
    assert a.shape[-1] == case["T"] * 2 * 16


@pytest.mark.parametrize("kind", ["snake", "snakebeta"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 1e-3)])
def test_aa_activation_matches_reference_torch_path(kind, dtype, tol):
    """the reference's own criterion for its CUDA kernel is mean-abs <= 1e-3 vs the torch path
    (BigVGAN/tests/test_activation.py:41); golden = the reference torch path on rand(10,10,200)."""
    from gsv.BigVGAN.alias_free_activation.cuda import anti_alias_activation as aa
    from gsv import synthetic as S
    g = load_golden("aa_" + kind)
    C_, T, B = 10, 200, 10
    x = torch.from_numpy(S.hash_uniform("aa_x_" + kind, B * C_ * T, 1).reshape(B, C_, T).copy())
    la = S.hash_symmetric("aa_alpha_" + kind, (C_,), 0.5, 1)
    lb = S.hash_symmetric("aa_beta_" + kind, (C_,), 0.5, 1) if kind == "snakebeta" else la
    uf = torch.from_numpy(g["up_filter"]).view(1, 1, 12)
    df = torch.from_numpy(g["down_filter"]).view(1, 1, 12)
    out = aa.forward(x.to(DEV, dtype), uf.to(DEV, dtype), df.to(DEV, dtype), la.to(DEV, dtype), lb.to(DEV, dtype))
    err = (out.float().cpu().numpy() - g["out"])
    assert np.abs(err).mean() <= tol
    assert np.abs(err).max() <= tol * 20


def test_aa_activation_shapes_and_edges():
    from gsv.BigVGAN.alias_free_activation.cuda import anti_alias_activation as aa
    from oracle import aa_oracle
    uf, df = aa_oracle.default_filters()
    for (B, C_, T) in [(1, 3, 1), (2, 5, 7), (1, 2, 2048), (1, 2, 2049), (3, 4, 5000)]:
        torch.manual_seed(T)
        x = torch.randn(B, C_, T)
        la, lb = torch.randn(C_) * 0.3, torch.randn(C_) * 0.3
        ref = aa_oracle.aa_activation(x, la, lb, uf, df)
        out = aa.forward(x.to(DEV), uf.view(1, 1, 12).to(DEV), df.view(1, 1, 12).to(DEV), la.to(DEV), lb.to(DEV))
        assert (out.cpu() - ref).abs().max() < 2e-5
    empty = aa.forward(torch.zeros(2, 3, 0, device=DEV), uf.view(1, 1, 12).to(DEV), df.view(1, 1, 12).to(DEV),
                       torch.zeros(3, device=DEV), torch.zeros(3, device=DEV))
    assert empty.shape == (2, 3, 0)


def _vocoder(cfg, sd, dtype):
    if cfg["kind"] == "hifigan":
        from gsv.module.models import Generator
        m = Generator(initial_channel=cfg["initial_channel"], resblock=cfg["resblock"],
                      resblock_kernel_sizes=cfg["resblock_kernel_sizes"],
                      resblock_dilation_sizes=cfg["resblock_dilation_sizes"], upsample_rates=cfg["upsample_rates"],
                      upsample_initial_channel=cfg["upsample_initial_channel"],
                      upsample_kernel_sizes=cfg["upsample_kernel_sizes"], gin_channels=0, is_bias=True, device=DEV, dtype=dtype)
    else:
        from gsv.BigVGAN.bigvgan import BigVGAN
        m = BigVGAN({k: v for k, v in cfg.items() if k != "kind"}, device=DEV, dtype=dtype)
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("name", ["voc_hifigan_small", "voc_bigvgan_small", "voc_hifigan_v4", "voc_bigvgan_v2"])
def test_vocoders_match_reference(name):
    """v4 HiFi-GAN vocoder (H16) and v3 BigVGAN (H15) vs the reference classes' output (golden):
    fp32 engine max-abs <= 2e-4; fp16 engine max-abs <= 3e-2 and relative RMS <= 5 %."""
    case = cases.VOC_CASES[name]
    cfg, sd, mel = cases.voc_case_inputs(case)
    g = load_golden(name)["wav"]
    out = _vocoder(cfg, sd, torch.float32)(mel.to(DEV)).float().cpu().numpy()
    assert out.shape == g.shape
    assert np.abs(out - g).max() <= 2e-4
    out16 = _vocoder(cfg, sd, torch.float16)(mel.to(DEV)).float().cpu().numpy()
    err = out16 - g
    assert np.abs(err).max() <= 3e-2
    assert np.sqrt((err ** 2).mean()) <= 0.05 * np.sqrt((g ** 2).mean())


def test_v2pro_conditioning_matches_reference():
    """v2Pro (N4): ge = mean_r PReLU(ref_enc(spec_r) + sv_emb(sv_r)) with gin 1024, the MRTE fed ge_to512(ge)
    (reference module/models.py:895-899, 971-975, 997), two references: fp32 engine vs the reference's waveform (golden)
    <= 1e-4, ge <= 1e-4; fp16 engine <= 2e-2; error behaviour of the sv_emb argument."""
    case = cases.VITS_CASES["vits_small_v2pro"]
    cfg, sd, codes, text, refers, noise, ssl = cases.vits_case_inputs(case)
    sv = cases.vits_case_sv_emb(case)
    g = load_golden("vits_small_v2pro")
    eng = _engine(cfg, sd, torch.float32)
    wav = eng.decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers], noise_scale=case["noise_scale"], noise=noise,
                     sv_emb=[v.to(DEV) for v in sv])
    ge = eng.debug_tensor("ge", 1024).cpu().numpy()
    assert np.abs(ge - g["ge"].reshape(-1)).max() < 1e-4
    assert np.abs(wav.float().cpu().numpy() - g["wav"]).max() <= 1e-4
    with pytest.raises(ValueError):
        eng.decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers], noise=noise)                 # sv_emb missing
    with pytest.raises(ValueError):
        eng.decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers], noise=noise, sv_emb=[sv[0].to(DEV)])
    w16 = _engine(cfg, sd, torch.float16).decode(codes.to(DEV), text.to(DEV), [r.to(DEV) for r in refers],
                                                 noise_scale=case["noise_scale"], noise=noise, sv_emb=sv).float().cpu().numpy()
    assert np.abs(w16 - g["wav"]).max() <= 2e-2
    v2 = _engine(*cases.vits_case_inputs(cases.VITS_CASES["vits_small"])[:2], torch.float32)
    with pytest.raises(ValueError):
        c2 = cases.vits_case_inputs(cases.VITS_CASES["vits_small"])
        v2.decode(c2[2].to(DEV), c2[3].to(DEV), [r.to(DEV) for r in c2[4]], sv_emb=sv[0])                  # not a v2Pro model


def test_fp32_folded_decode_at_benchmark_length_vs_oracle():
    """BASELINE configs[1] SoVITS shape at engine level: two utterances of 100 tokens folded into the time axis (400 frames,
    256 000 samples, enc_p attention across both sentences) through the full v2 architecture, fp32 engine vs the CPU oracle
    (pinned against the reference class on vits_v2): waveform max-abs error <= 1e-4."""
    from gsv import synthetic as S
    from gsv.module.models import SynthesizerTrn
    from oracle.vits_oracle import VitsOracle
    cfg = S.VITS_V2_CONFIG
    sd = S.make_vits_state_dict(cfg, seed=0)
    d = cfg["data"]
    v = SynthesizerTrn(d["filter_length"] // 2 + 1, cfg["train"]["segment_size"] // d["hop_length"], n_speakers=d["n_speakers"],
                       version="v2", device="cuda:0", dtype=torch.float32, n_symbols=cfg.get("n_symbols"), **cfg["model"])
    v.load_state_dict(sd)
    codes = torch.from_numpy(S.hash_ints("bench_codes", 200, 1024, 0)).view(1, 1, -1)
    text = torch.from_numpy(S.hash_ints("bench_text", 80, 732, 0)).view(1, -1)
    refer = S.make_refer_spec()
    noise = S.hash_normal("bench_vits_noise", (cfg["model"]["inter_channels"], 400), 0)
    torch.set_num_threads(8)
    ref = VitsOracle(sd, cfg).decode(codes, text, [refer], noise=noise)
    wav = v.decode(codes.to("cuda:0"), text.to("cuda:0"), [refer.to("cuda:0")], noise=noise).float().cpu()
    assert wav.shape == ref.shape == (1, 1, 400 * 640)
    err = (wav - ref).abs().max().item()
    print(f"[parity] fp32 SoVITS folded decode, 400 frames: max-abs error {err:.2e} (waveform rms {ref.pow(2).mean().sqrt():.3f})")
    assert err <= 1e-4


def test_fp16_folded_decode_at_benchmark_length_vs_oracle():
    """The production dtype at the benchmark's SoVITS shape: two 100-token utterances folded into the time
    axis (400 frames, 256 000 samples).  At this length the decode runs the fp16-only kernels the bench spends its time in --
    `conv_wide_f16` (128 channels, T = 32 000 >= 16 384), `conv_pair_f16` (32 / 16 channels), `conv_narrow_f16` (64 channels),
    `flash_rel96_f16` (enc_p attention) -- composed end to end, against the fp32 CPU oracle (pinned on the reference class):
    waveform max-abs <= 2e-2 and relative rms <= 3 %, the fp16 bar of DESIGN.md section 2."""
    from gsv import synthetic as S
    from gsv.module.models import SynthesizerTrn
    from oracle.vits_oracle import VitsOracle
    cfg = S.VITS_V2_CONFIG
    sd = S.make_vits_state_dict(cfg, seed=0)
    d = cfg["data"]
    v = SynthesizerTrn(d["filter_length"] // 2 + 1, cfg["train"]["segment_size"] // d["hop_length"], n_speakers=d["n_speakers"],
                       version="v2", device="cuda:0", dtype=torch.float16, n_symbols=cfg.get("n_symbols"), **cfg["model"])
    v.load_state_dict(sd)
    codes = torch.from_numpy(S.hash_ints("bench_codes", 200, 1024, 0)).view(1, 1, -1)
    text = torch.from_numpy(S.hash_ints("bench_text", 80, 732, 0)).view(1, -1)
    refer = S.make_refer_spec()
    noise = S.hash_normal("bench_vits_noise", (cfg["model"]["inter_channels"], 400), 0)
    torch.set_num_threads(8)
    ref = VitsOracle(sd, cfg).decode(codes, text, [refer], noise=noise)
    wav = v.decode(codes.to("cuda:0"), text.to("cuda:0"), [refer.to("cuda:0")], noise=noise).float().cpu()
    assert wav.shape == ref.shape == (1, 1, 400 * 640)
    assert torch.isfinite(wav).all()
    err = (wav - ref).abs().max().item()
    rel = ((wav - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"[parity] fp16 SoVITS folded decode, 400 frames: max-abs error {err:.2e}, relative rms {rel * 100:.2f} % "
          f"(waveform rms {ref.pow(2).mean().sqrt():.3f})")
    assert err <= 2e-2 and rel <= 0.03
