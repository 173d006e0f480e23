"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32 on the host) of the speaker-verification embedding the v2Pro / v2ProPlus
models are conditioned on: `SV.compute_embedding3` (reference GPT_SoVITS/sv.py:24-32) =
Kaldi fbank (eres2net/kaldi.py:519-676 with num_mel_bins=80, sample_frequency=16000, dither=0, everything else default)
-> ERes2NetV2(baseWidth=24, scale=4, expansion=4).forward3 (eres2net/ERes2NetV2.py:246-258, blocks :28-151, AFF
eres2net/fusion.py:8-27).  Pinned by tests/golden/sv_eres2net.npz, which holds outputs of the reference's own classes run in the
build container on gsv.synthetic weights (oracle/gen_golden_frontend.py)."""
import math

import torch
import torch.nn.functional as F

FLT_EPS = 1.1920928955078125e-07


def mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0):
    """kaldi.py:436-513 without VTLN: triangles equally spaced on the HTK mel scale 1127 ln(1 + f / 700) -> [num_bins][padded/2]"""
    nyq = 0.5 * sr
    high = high + nyq if high <= 0 else high
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(low), mel(high)
    delta = (hi - lo) / (num_bins + 1)
    out = torch.zeros(num_bins, padded // 2)
    width = sr / padded
    for b in range(num_bins):
        left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
        for k in range(padded // 2):
            m = mel(width * k)
            out[b, k] = max(0.0, min((m - left) / (center - left), (right - m) / (right - center)))
    return out


def fbank(wav: torch.Tensor, num_mel_bins=80, sr=16000):
    """wav [n] fp32 -> [m, num_mel_bins]: 25 ms frames every 10 ms (snip_edges), DC removal, pre-emphasis 0.97 with the first
    sample replicated, Povey window (Hann^0.85, symmetric), zero padding to 512, power spectrum, mel banks, log(max(., eps))"""
    win, shift, padded = int(sr * 0.025), int(sr * 0.010), 512
    n = wav.shape[0]
    m = 1 + (n - win) // shift
    frames = torch.stack([wav[i * shift:i * shift + win] for i in range(m)])
    frames = frames - frames.mean(1, keepdim=True)
    prev = torch.cat([frames[:, :1], frames[:, :-1]], 1)
    frames = frames - 0.97 * prev
    frames = frames * torch.hann_window(win, periodic=False).pow(0.85)
    frames = F.pad(frames, (0, padded - win))
    spec = torch.fft.rfft(frames).abs().pow(2.0)                                  # [m, 257]
    banks = F.pad(mel_banks(num_mel_bins, padded, float(sr)), (0, 1))              # [80, 257]
    return torch.clamp(spec @ banks.T, min=FLT_EPS).log()


class ERes2NetV2Oracle:
    def __init__(self, sd, base_width=24, scale=4, expansion=4, m_channels=64, num_blocks=(3, 4, 6, 3)):
        self.sd = {k: v.float() for k, v in sd.items()}
        self.scale, self.expansion = scale, expansion
        self.plan = []
        in_planes = m_channels
        for li, (planes, nb, stride) in enumerate(zip((m_channels, 2 * m_channels, 4 * m_channels, 8 * m_channels), num_blocks, (1, 2, 2, 2)), 1):
            width = int(math.floor(planes * (base_width / 64.0)))
            for bi in range(nb):
                st = stride if bi == 0 else 1
                self.plan.append((f"layer{li}.{bi}", st, width, li >= 3, st != 1 or in_planes != planes * expansion))
                in_planes = planes * expansion

    def bn(self, x, p):
        s = self.sd
        return F.batch_norm(x, s[p + ".running_mean"], s[p + ".running_var"], s[p + ".weight"], s[p + ".bias"], False, 0.0, 1e-5)

    def aff(self, p, x, y):
        s = self.sd
        h = F.conv2d(torch.cat([x, y], 1), s[p + ".local_att.0.weight"], s[p + ".local_att.0.bias"])
        h = F.silu(self.bn(h, p + ".local_att.1"))
        h = self.bn(F.conv2d(h, s[p + ".local_att.3.weight"], s[p + ".local_att.3.bias"]), p + ".local_att.4")
        att = 1.0 + torch.tanh(h)
        return x * att + y * (2.0 - att)

    def block(self, p, x, stride, width, fuse, has_sc):
        s = self.sd
        relu20 = lambda v: torch.clamp(v, 0.0, 20.0)
        out = relu20(self.bn(F.conv2d(x, s[p + ".conv1.weight"], stride=stride), p + ".bn1"))
        spx = torch.split(out, width, 1)
        outs, sp = [], None
        for i in range(self.scale):
            if i == 0:
                sp = spx[0]
            elif fuse:
                sp = self.aff(p + f".fuse_models.{i - 1}", sp, spx[i])
            else:
                sp = sp + spx[i]
            sp = relu20(self.bn(F.conv2d(sp, s[p + f".convs.{i}.weight"], padding=1), p + f".bns.{i}"))
            outs.append(sp)
        out = self.bn(F.conv2d(torch.cat(outs, 1), s[p + ".conv3.weight"]), p + ".bn3")
        res = self.bn(F.conv2d(x, s[p + ".shortcut.0.weight"], stride=stride), p + ".shortcut.1") if has_sc else x
        return relu20(out + res)

    def forward3(self, feat: torch.Tensor, taps=None):
        """feat [B, T, 80] -> [B, 20480] (channel-major, then the 10 frequency rows), mean over time"""
        s = self.sd
        x = feat.permute(0, 2, 1).unsqueeze(1)
        out = F.relu(self.bn(F.conv2d(x, s["conv1.weight"], padding=1), "bn1"))
        out3 = None
        for p, st, width, fuse, has_sc in self.plan:
            out = self.block(p, out, st, width, fuse, has_sc)
            if taps is not None:
                taps[p] = out
            if p.startswith("layer3.") and not any(q[0] > p and q[0].startswith("layer3.") for q in self.plan):
                out3 = out
        out3_ds = F.conv2d(out3, s["layer3_ds.weight"], stride=2, padding=1)
        fused = self.aff("fuse34", out, out3_ds)
        return fused.flatten(1, 2).mean(-1)


def compute_embedding3(sd, wav: torch.Tensor) -> torch.Tensor:
    """wav [B, n] fp32 at 16 kHz -> [B, 20480]"""
    feat = torch.stack([fbank(w) for w in wav])
    return ERes2NetV2Oracle(sd).forward3(feat)
