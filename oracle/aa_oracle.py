"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of BigVGAN's anti-aliased
snake activation, the op the reference's single native kernel fuses
(reference BigVGAN/alias_free_activation/cuda/anti_alias_activation_cuda.cu:44-179;
torch path alias_free_activation/torch/{act.py:25-30, resample.py:23-30,45-48,
filter.py:30-60}; Snake/SnakeBeta BigVGAN/activations.py:9-122).  Never imported by
the product path.  Pinned by oracle/gen_golden_vits.py against the reference's own
torch path (the CUDA kernel cannot run here); the reference's own tolerance for
kernel-vs-torch is mean-abs <= 1e-3 (BigVGAN/tests/test_activation.py:41).
"""
import math

import torch
import torch.nn.functional as F


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """Kaiser-windowed sinc low-pass, normalised to unit DC gain (filter.py:30-60)."""
    half = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if kernel_size % 2 == 0:
        t = torch.arange(-half, half) + 0.5
    else:
        t = torch.arange(kernel_size) - half
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * t)
    return f / f.sum()


def default_filters():
    """The 12-tap up/down filters Activation1d builds for ratio 2 (resample.py:17,41-43)."""
    f = kaiser_sinc_filter1d(0.25, 0.3, 12)
    return f.clone(), f.clone()


def aa_activation(x: torch.Tensor, log_alpha: torch.Tensor, log_beta: torch.Tensor,
                  up_f: torch.Tensor, down_f: torch.Tensor) -> torch.Tensor:
    """x [B, C, T]; log-scale alpha/beta [C] (what the kernel receives, activation1d.py:58-66).
    2x upsample (replicate pad 5, zero-stuffed 12-tap FIR, gain 2) -> x + sin^2(a x)/(b+1e-9)
    -> replicate pad (5, 6) -> 12-tap FIR stride 2."""
    B, C, T = x.shape
    x = x.float()
    xp = F.pad(x, (5, 5), mode="replicate")
    up = 2.0 * F.conv_transpose1d(xp, up_f.view(1, 1, 12).expand(C, -1, -1).contiguous(), stride=2, groups=C)
    up = up[..., 15:-15]                                   # pad_left = 5*2+5, pad_right = 5*2+5
    a = torch.exp(log_alpha.float()).view(1, C, 1)
    b = torch.exp(log_beta.float()).view(1, C, 1)
    act = up + (1.0 / (b + 1e-9)) * torch.sin(up * a) ** 2
    ap = F.pad(act, (5, 6), mode="replicate")
    return F.conv1d(ap, down_f.view(1, 1, 12).expand(C, -1, -1).contiguous(), stride=2, groups=C)
