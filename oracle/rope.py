"""TEST INFRASTRUCTURE ONLY -- restatement of x_transformers' rotary embedding (the reference imports
`RotaryEmbedding` and `apply_rotary_pos_emb` from the un-vendored, un-pinned `x_transformers`,
f5_tts/model/backbones/dit.py:16, f5_tts/model/modules.py:24).  PARITY UNPINNED: the library is not in
the container, so this follows its published definition (x_transformers >= 1.31: interleaved pairs,
`freqs = stack((f, f), -1)` flattened, `rotate_half` on adjacent pairs, partial rotary = only the first
`freqs.shape[-1]` channels of the un-split [b, n, heads*dim_head] projection are rotated).
"""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, use_xpos=False, scale_base=512, interpolation_factor=1.0, base=10000, base_rescale_factor=1.0):
        super().__init__()
        base = base * base_rescale_factor ** (dim / (dim - 2))
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)), persistent=False)
        self.interpolation_factor = interpolation_factor

    def forward_from_seq_len(self, seq_len):
        return self.forward(torch.arange(seq_len, device=self.inv_freq.device))

    def forward(self, t):
        if t.ndim == 1:
            t = t.unsqueeze(0)
        freqs = torch.einsum("bi,j->bij", t.type_as(self.inv_freq), self.inv_freq) / self.interpolation_factor
        freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)          # '... d r -> ... (d r)'
        return freqs, 1.0


def rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[:, -seq_len:, :]
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs.unsqueeze(1)
    t_rot, t_unrot = t[..., :rot_dim], t[..., rot_dim:]
    t_rot = (t_rot * freqs.cos() * scale) + (rotate_half(t_rot) * freqs.sin() * scale)
    return torch.cat((t_rot, t_unrot), dim=-1).type(orig_dtype)
