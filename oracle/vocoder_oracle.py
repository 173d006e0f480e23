"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the v3/v4 vocoders: the HiFi-GAN `Generator`
used as a mel vocoder for v4 (reference module/models.py:407-471 via TTS_infer_pack/TTS.py:631-648) and
BigVGAN-v2 for v3 (reference BigVGAN/bigvgan.py:31-131, 226-355).  Never imported by the product path.
Pinned against the reference classes by oracle/gen_golden_vits.py -> tests/golden/voc_*.npz."""
import torch
import torch.nn.functional as F

from . import aa_oracle


def hifigan(sd, cfg, mel):
    """mel [1, 100, F] -> [1, 1, F*prod(rates)]"""
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = f"resblocks.{i * nk + j}"
            xr = x
            for c, d in enumerate(rd):
                xt = F.conv1d(F.leaky_relu(xr, 0.1), sd[f"{r}.convs1.{c}.weight"], sd[f"{r}.convs1.{c}.bias"],
                              padding=(rk * d - d) // 2, dilation=d)
                xt = F.conv1d(F.leaky_relu(xt, 0.1), sd[f"{r}.convs2.{c}.weight"], sd[f"{r}.convs2.{c}.bias"],
                              padding=(rk - 1) // 2)
                xr = xt + xr
            xs = xr if xs is None else xs + xr
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], sd.get("conv_post.bias"), padding=3)
    return torch.tanh(x)


def bigvgan(sd, cfg, mel):
    """mel [1, 100, F] -> [1, 1, F*prod(rates)]; anti-aliased SnakeBeta before every conv."""
    uf, df = aa_oracle.default_filters()

    def act(x, prefix):
        a, b = sd[prefix + ".alpha"], sd[prefix + ".beta"]
        if not cfg.get("snake_logscale", True):
            a, b = torch.log(a), torch.log(b)
        return aa_oracle.aa_activation(x, a, b, uf, df)

    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, sd[f"ups.{i}.0.weight"], sd[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = f"resblocks.{i * nk + j}"
            xr = x
            for c, d in enumerate(rd):
                xt = act(xr, f"{r}.activations.{2 * c}.act")
                xt = F.conv1d(xt, sd[f"{r}.convs1.{c}.weight"], sd[f"{r}.convs1.{c}.bias"], padding=(rk * d - d) // 2, dilation=d)
                xt = act(xt, f"{r}.activations.{2 * c + 1}.act")
                xt = F.conv1d(xt, sd[f"{r}.convs2.{c}.weight"], sd[f"{r}.convs2.{c}.bias"], padding=(rk - 1) // 2)
                xr = xt + xr
            xs = xr if xs is None else xs + xr
        x = xs / nk
    x = act(x, "activation_post.act")
    x = F.conv1d(x, sd["conv_post.weight"], sd.get("conv_post.bias"), padding=3)
    return torch.tanh(x) if cfg.get("use_tanh_at_final", True) else torch.clamp(x, -1.0, 1.0)
