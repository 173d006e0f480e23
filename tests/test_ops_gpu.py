"""GPU: single kernels through the C ABI against plain torch fp32 references of the same op."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _conv(x_ct, w, bias, dtype, dil=1, stride=1, pre_lrelu=None, res=None, post=0, scale=1.0, accumulate=None):
    """x_ct [Cin, T] fp32 cpu, w [Cout, Cin, k] -> y [Cout, T_out] via gsv_op_conv1d (channels-last inside)."""
    from gsv import _lib
    _lib.init(0)
    Cout, Cin, k = w.shape
    T = x_ct.shape[1]
    pad = (k * dil - dil) // 2
    T_out = (T + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = x_ct.t().contiguous().to(DEV, dtype)
    wp = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous().to(DEV, dtype)
    y = torch.zeros(T_out, Cout, device=DEV, dtype=dtype) if accumulate is None else accumulate.t().contiguous().to(DEV, dtype)
    b = bias.to(DEV, torch.float32).contiguous() if bias is not None else None
    r = res.t().contiguous().to(DEV, dtype) if res is not None else None
    d = _lib.ConvDesc(x.data_ptr(), wp.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                      r.data_ptr() if r is not None else None, T, T_out, Cin, Cout, k, stride, dil, pad,
                      3 if pre_lrelu is not None else 0, pre_lrelu or 0.0, post, scale,
                      1 if accumulate is not None else 0, 0, 0, 0)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.dtype_code(dtype), None))
    torch.cuda.synchronize()
    return y.float().cpu().t()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-2)])
@pytest.mark.parametrize("Cin,Cout,k,dil,T", [(192, 512, 7, 1, 50), (256, 256, 11, 5, 333), (64, 64, 3, 3, 700),
                                               (16, 16, 11, 1, 1000), (32, 32, 7, 5, 129), (768, 192, 1, 1, 37),
                                               (96, 40, 3, 1, 65), (8, 24, 5, 1, 31)])
def test_conv1d_matches_torch(dtype, tol, Cin, Cout, k, dil, T):
    torch.manual_seed(Cin * 31 + k)
    x = torch.randn(Cin, T)
    w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
    b = torch.randn(Cout)
    res = torch.randn(Cout, T)
    ref = F.conv1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, padding=(k * dil - dil) // 2, dilation=dil)[0] + res
    got = _conv(x, w, b, dtype, dil=dil, pre_lrelu=0.1, res=res)
    assert (got - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_conv1d_edges_stride_scale_accumulate_tanh():
    torch.manual_seed(3)
    x = torch.randn(24, 41)
    w = torch.randn(16, 24, 2) / 7
    ref = F.conv1d(x.unsqueeze(0), w, None, stride=2)[0]
    from gsv import _lib
    # stride-2, k=2, no padding (the ssl_proj shape of extract_latent)
    _lib.init(0)
    xd = x.t().contiguous().to(DEV)
    wp = w.permute(0, 2, 1).reshape(16, 48).contiguous().to(DEV)
    y = torch.zeros(20, 16, device=DEV)
    d = _lib.ConvDesc(xd.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 41, 20, 24, 16, 2, 2, 1, 0, 0, 0.0, 0,
                      1.0, 0, 0, 0, 0)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 0, None))
    torch.cuda.synchronize()
    assert (y.cpu().t() - ref).abs().max() < 1e-5
    # scale + accumulate + tanh
    x2 = torch.randn(32, 77)
    w2 = torch.randn(32, 32, 3) / 10
    prev = torch.randn(32, 77)
    ref2 = prev + torch.tanh(F.conv1d(x2.unsqueeze(0), w2, None, padding=1)[0] / 3.0)
    got2 = _conv(x2, w2, None, torch.float32, post=2, scale=1.0 / 3.0, accumulate=prev)
    assert (got2 - ref2).abs().max() < 1e-5


@pytest.mark.parametrize("u,k,Cin,Cout,T", [(10, 16, 64, 32, 23), (8, 16, 32, 16, 40), (2, 8, 128, 64, 33),
                                             (2, 2, 64, 32, 50), (4, 8, 128, 64, 17)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-2)])
def test_transposed_conv_as_polyphase_conv(u, k, Cin, Cout, T, dtype, tol):
    """ConvTranspose1d(stride u, padding (k-u)/2) restated as a ceil(k/u)-tap conv producing u*Cout
    virtual channels with a scatter epilogue (the generator's upsamplers, models.py:427-437)."""
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(u * 7 + k)
    x = torch.randn(Cin, T)
    w = torch.randn(Cin, Cout, k) / (Cin * k / u) ** 0.5
    b = torch.randn(Cout)
    pad = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, stride=u, padding=pad)[0]
    taps = -(-k // u)
    wv = torch.zeros(u * Cout, taps, Cin)
    for p in range(u):
        for q in range(taps):
            j = q * u + p
            if j < k:
                wv[p * Cout:(p + 1) * Cout, q, :] = w[:, :, j].t()
    xd = x.t().contiguous().to(DEV, dtype)
    wd = wv.reshape(u * Cout, taps * Cin).contiguous().to(DEV, dtype)
    bd = b.to(DEV)
    T_out = T * u
    y = torch.zeros(T_out, Cout, device=DEV, dtype=dtype)
    d = _lib.ConvDesc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), None, T, T_out, Cin, u * Cout, taps, 1,
                      -1, 0, 3, 0.1, 0, 1.0, 0, 0, u, pad)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.dtype_code(dtype), None))
    torch.cuda.synchronize()
    assert ref.shape[1] == T_out
    assert (y.float().cpu().t() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_layernorm_matches_torch():
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(1)
    for dtype, tol in [(torch.float32, 1e-5), (torch.float16, 4e-3)]:
        x = torch.randn(37, 512)
        r = torch.randn(37, 512)
        g, b = torch.rand(512) + 0.5, torch.randn(512)
        ref = F.layer_norm((x.to(dtype) + r.to(dtype)).float(), [512], g, b, 1e-5)
        xd, rd, gd, bd = x.to(DEV, dtype), r.to(DEV, dtype), g.to(DEV), b.to(DEV)
        y = torch.empty_like(xd)
        _lib.check(_lib.lib().gsv_op_layernorm(xd.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                               y.data_ptr(), 37, 512, 1e-5, _lib.dtype_code(dtype), None))
        torch.cuda.synchronize()
        assert (y.float().cpu() - ref).abs().max() < tol * 10


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.float16, 2e-2)])
@pytest.mark.parametrize("Cin,Cout,k,dil,T", [(256, 256, 11, 5, 700), (192, 512, 7, 1, 300), (128, 128, 3, 3, 1111),
                                               (64, 64, 7, 5, 2000), (32, 32, 11, 1, 1500), (16, 16, 3, 5, 4100),
                                               (16, 1, 7, 1, 3000), (768, 192, 3, 1, 260), (192, 384, 5, 1, 513),
                                               (96, 192, 1, 1, 400), (512, 1536, 1, 1, 700), (2048, 512, 1, 1, 600),
                                               (192, 576, 1, 1, 1000), (704, 128, 1, 1, 520)])
def test_conv1d_lds_path_matches_torch(dtype, tol, Cin, Cout, k, dil, T):
    """T >= 256 and stride 1 route through the LDS-staged kernel (conv_lds.hip): every tile config,
    partial channel chunks (192 = 128 + 64), halo rows outside the sequence, Cout = 1 (conv_post)."""
    torch.manual_seed(Cin + Cout + k)
    x = torch.randn(Cin, T)
    w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
    b = torch.randn(Cout)
    res = torch.randn(Cout, T)
    ref = F.conv1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, padding=(k * dil - dil) // 2, dilation=dil)[0] + res
    got = _conv(x, w, b, dtype, dil=dil, pre_lrelu=0.1, res=res)
    assert (got - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    # scale + accumulate (the MRF mean of the generator, models.py:459-466)
    prev = torch.randn(Cout, T)
    ref2 = prev + (F.conv1d(x.unsqueeze(0), w, b, padding=(k * dil - dil) // 2, dilation=dil)[0] + res) / 3.0
    got2 = _conv(x, w, b, dtype, dil=dil, res=res, scale=1.0 / 3.0, accumulate=prev)
    assert (got2 - ref2).abs().max().item() < tol * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("u,k,Cin,Cout,T", [(10, 16, 512, 256, 300), (8, 16, 256, 128, 600), (2, 8, 128, 64, 1000),
                                             (2, 2, 64, 32, 2500), (2, 2, 32, 16, 3000)])
def test_transposed_conv_lds_path(u, k, Cin, Cout, T):
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(u + k + Cin)
    dtype = torch.float16
    x = torch.randn(Cin, T)
    w = torch.randn(Cin, Cout, k) / (Cin * k / u) ** 0.5
    b = torch.randn(Cout)
    pad = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, stride=u, padding=pad)[0]
    taps = -(-k // u)
    wv = torch.zeros(u * Cout, taps, Cin)
    for p in range(u):
        for q in range(taps):
            j = q * u + p
            if j < k:
                wv[p * Cout:(p + 1) * Cout, q, :] = w[:, :, j].t()
    xd = x.t().contiguous().to(DEV, dtype)
    wd = wv.reshape(u * Cout, taps * Cin).contiguous().to(DEV, dtype)
    bd = b.to(DEV)
    T_out = T * u
    y = torch.zeros(T_out, Cout, device=DEV, dtype=dtype)
    d = _lib.ConvDesc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), None, T, T_out, Cin, u * Cout, taps, 1,
                      -1, 0, 3, 0.1, 0, 1.0, 0, 0, u, pad)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.dtype_code(dtype), None))
    torch.cuda.synchronize()
    assert (y.float().cpu().t() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.float16, 2e-2)])
@pytest.mark.parametrize("Tq,Tk,nh,kc", [(700, 650, 2, 96), (1300, 520, 4, 128)])
def test_batched_attention_gemms(dtype, tol, Tq, Tk, nh, kc):
    """the two attention products of enc_p as batched GEMMs over heads (Z = heads, strided head slices):
    scores = scale * Q K^T (fp32 out) and out = P V^T-layout, both through the LDS GEMM kernel."""
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(Tq + Tk)
    H = nh * kc
    q = torch.randn(Tq, H)
    k = torch.randn(Tk, H)
    qd, kd = q.to(DEV, dtype).contiguous(), k.to(DEV, dtype).contiguous()
    sc = torch.zeros(nh, Tq, Tk, device=DEV)
    d = _lib.ConvDesc(qd.data_ptr(), kd.data_ptr(), None, sc.data_ptr(), None, Tq, Tq, kc, Tk, 1, 1, 1, 0, 0, 0.0, 0, 0.125, 0,
                      1, 0, 0, nh, kc, kc, Tq * Tk, H, H, Tk)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), _lib.dtype_code(dtype), None))
    torch.cuda.synchronize()
    ref = torch.einsum("qhc,khc->hqk", q.view(Tq, nh, kc), k.view(Tk, nh, kc)) * 0.125
    assert (sc.cpu() - ref).abs().max().item() < tol * ref.abs().max().item()
    # P [nh][Tq][Tkp] x Vt [nh][kc][Tkp] -> out [Tq][H]
    Tkp = (Tk + 7) // 8 * 8
    P = torch.zeros(nh, Tq, Tkp)
    P[:, :, :Tk] = torch.softmax(ref, dim=-1)
    v = torch.randn(Tk, H)
    Vt = torch.zeros(nh, kc, Tkp)
    Vt[:, :, :Tk] = v.view(Tk, nh, kc).permute(1, 2, 0)
    Pd, Vd = P.to(DEV, dtype).contiguous(), Vt.to(DEV, dtype).contiguous()
    out = torch.zeros(Tq, H, device=DEV, dtype=dtype)
    d2 = _lib.ConvDesc(Pd.data_ptr(), Vd.data_ptr(), None, out.data_ptr(), None, Tq, Tq, Tkp, kc, 1, 1, 1, 0, 0, 0.0, 0, 1.0, 0,
                       0, 0, 0, nh, Tq * Tkp, kc * Tkp, kc, Tkp, Tkp, H)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d2), _lib.dtype_code(dtype), None))
    torch.cuda.synchronize()
    ref2 = torch.einsum("hqk,khc->qhc", P[:, :, :Tk], v.view(Tk, nh, kc)).reshape(Tq, H)
    assert (out.float().cpu() - ref2).abs().max().item() < tol * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("T,heads", [(934, 16), (33, 2), (16, 1), (1, 1), (129, 3)])
def test_flash_attention_matches_sdpa(T, heads):
    """fused DiT attention (fp16, d=64) vs torch fp32 softmax(QK^T/8)V on the same fp16-rounded inputs: max-abs <= 4e-3
    (fp16 probabilities / output rounding), including ragged T (partial key chunks, partial query tiles)."""
    from gsv import _lib
    from gsv import synthetic as S
    _lib.init(0)
    inner = heads * 64
    qkv = (S.hash_symmetric(f"fa_qkv_{T}", (T, 3 * inner), 1.5, 2)).to(DEV, torch.float16)
    vt = torch.empty(heads * 64 * ((T + 31) // 32 * 32), dtype=torch.float16, device=DEV)
    out = torch.empty(T, inner, dtype=torch.float16, device=DEV)
    st = torch.cuda.current_stream()
    _lib.check(_lib.lib().gsv_op_flash_attn64(qkv.data_ptr(), T, heads, 0.125, vt.data_ptr(), out.data_ptr(), C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    x = qkv.float().cpu()
    q, k, v = [x[:, i * inner:(i + 1) * inner].view(T, heads, 64).transpose(0, 1) for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, -1) @ v).transpose(0, 1).reshape(T, inner)
    assert (out.float().cpu() - ref).abs().max() <= 4e-3


@pytest.mark.parametrize("T,K,N,post,use_res,use_gate", [(934, 1024, 3072, 0, False, False), (934, 2048, 1024, 0, True, True),
                                                          (100, 1024, 2048, 8, False, False), (77, 320, 200, 6, True, False),
                                                          (16, 256, 64, 0, False, True)])
def test_skinny_gemm_matches_torch(T, K, N, post, use_res, use_gate):
    """split-K-in-workgroup streaming GEMM (gemm_sk.hip: Linear layers whose grid cannot fill the chip), fp16 operands,
    fp32 accumulation, gate / residual / GELU-tanh / SiLU epilogues, ragged T and N: <= 2e-2 relative to the output range."""
    from gsv import _lib
    from gsv import synthetic as S
    _lib.init(0)
    x = S.hash_symmetric(f"sk_x{T}", (T, K), 1.0, 1).to(DEV, torch.float16)
    w = (S.hash_symmetric(f"sk_w{N}", (N, K), 1.0, 1) / K ** 0.5 * 3).to(DEV, torch.float16)
    b = S.hash_symmetric("sk_b", (N,), 0.5, 1).to(DEV)
    gate = S.hash_symmetric("sk_g", (N,), 1.0, 1).to(DEV) if use_gate else None
    res = S.hash_symmetric("sk_r", (T, N), 1.0, 1).to(DEV, torch.float16) if use_res else None
    y = torch.zeros(T, N, device=DEV, dtype=torch.float16)
    d = _lib.ConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), res.data_ptr() if use_res else None, T, T, K, N, 1, 1, 1,
                      0, 0, 0.0, post, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, gate.data_ptr() if use_gate else None)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 1, None))
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + b
    if use_gate:
        ref = ref * gate
    if use_res:
        ref = ref + res.float()
    if post == 8:
        ref = F.gelu(ref, approximate="tanh")
    elif post == 6:
        ref = F.silu(ref)
    assert (y.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("T,heads", [(6400, 2), (37, 2), (5, 1), (1, 2), (200, 3)])
def test_flash_rel96_matches_reference_formula(T, heads):
    """fused enc_p attention (fp16, d = 96, window-4 relative keys / values) vs the reference's formula in torch fp32
    (module/attentions.py:227-258: scores + q.rel_k on the band, softmax, p @ v + band(p) @ rel_v) on the same
    fp16-rounded inputs: max-abs <= 5e-3."""
    from gsv import _lib
    from gsv import synthetic as S
    _lib.init(0)
    D, W = 96, 4
    H = heads * D
    qkv = (S.hash_symmetric(f"rel_qkv_{T}", (T, 3 * H), 1.2, 3)).to(DEV, torch.float16)
    rel_k = S.hash_symmetric("rel_k", (2 * W + 1, D), 0.3, 3).to(DEV)
    rel_v = S.hash_symmetric("rel_v", (2 * W + 1, D), 0.5, 3).to(DEV)
    vt = torch.empty(heads * D * ((T + 31) // 32 * 32), dtype=torch.float16, device=DEV)
    out = torch.empty(T, H, dtype=torch.float16, device=DEV)
    scale = 1.0 / D ** 0.5
    st = torch.cuda.current_stream()
    _lib.check(_lib.lib().gsv_op_flash_rel96(qkv.data_ptr(), T, heads, scale, rel_k.data_ptr(), rel_v.data_ptr(), vt.data_ptr(),
                                             out.data_ptr(), C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    x = qkv.float()
    rk, rv = rel_k, rel_v
    ref = torch.empty(T, H, device=DEV)
    idx = torch.arange(T, device=DEV)
    rel = idx[None, :] - idx[:, None] + W                               # [T, T] band index, valid in [0, 2W]
    inband = (rel >= 0) & (rel <= 2 * W)
    for h in range(heads):
        q = x[:, h * D:(h + 1) * D] * scale
        k = x[:, H + h * D:H + (h + 1) * D]
        v = x[:, 2 * H + h * D:2 * H + (h + 1) * D]
        s = q @ k.t()
        b = q @ rk.t()                                                  # [T, 9]
        s = s + torch.where(inband, b.gather(1, rel.clamp(0, 2 * W)), torch.zeros((), device=DEV))
        p = torch.softmax(s, -1)
        o = p @ v
        pb = torch.zeros(T, 2 * W + 1, device=DEV)
        pb.scatter_add_(1, rel.clamp(0, 2 * W), torch.where(inband, p, torch.zeros((), device=DEV)))
        ref[:, h * D:(h + 1) * D] = o + pb @ rv
    assert (out.float() - ref).abs().max().item() <= 5e-3


@pytest.mark.parametrize("C,k,dil,T", [(16, 11, 1, 5000), (16, 3, 5, 8192), (32, 7, 3, 4500), (32, 11, 5, 4096 + 77),
                                       (64, 11, 5, 16384 + 100)])
@pytest.mark.parametrize("mode", ["plain", "res", "res_acc"])
def test_persistent_narrow_conv_matches_torch(C, k, dil, T, mode):
    """persistent narrow-layer kernel (conv_narrow_f16_kernel: C = 16 / 32 / 64, long T, several tiles per workgroup with the
    next window prefetched): leaky-relu on load, bias, residual, scale + accumulate, ragged last tile; fp16 vs torch fp32."""
    torch.manual_seed(C + k + dil)
    x = torch.randn(C, T)
    w = torch.randn(C, C, k) / (C * k) ** 0.5
    b = torch.randn(C)
    res = torch.randn(C, T) if mode != "plain" else None
    prev = torch.randn(C, T) if mode == "res_acc" else None
    conv = F.conv1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, padding=(k * dil - dil) // 2, dilation=dil)[0]
    ref = conv + (res if res is not None else 0)
    if prev is not None:
        ref = prev + ref / 3.0
    got = _conv(x, w, b, torch.float16, dil=dil, pre_lrelu=0.1, res=res, scale=(1.0 / 3.0 if prev is not None else 1.0),
                accumulate=prev)
    assert (got - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


def test_persistent_narrow_conv_single_fp32_output_channel():
    """the generator's conv_post shape on the persistent kernel: 16 -> 1 channels, k = 7, leaky-relu(0.01) on load, tanh,
    fp32 output (reference module/models.py:467-470)."""
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(5)
    Cin, k, T = 16, 7, 6000
    x = torch.randn(Cin, T)
    w = torch.randn(1, Cin, k) / (Cin * k) ** 0.5
    ref = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01).unsqueeze(0), w, None, padding=3)[0])
    xd = x.t().contiguous().to(DEV, torch.float16)
    wp = w.permute(0, 2, 1).reshape(1, k * Cin).contiguous().to(DEV, torch.float16)
    y = torch.zeros(T, 1, device=DEV, dtype=torch.float32)
    d = _lib.ConvDesc(xd.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, T, T, Cin, 1, k, 1, 1, 3, 3, 0.01, 2, 1.0, 0, 1, 0, 0,
                      0, 0, 0, 0, 0, 0, 0, None)
    _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 1, None))
    torch.cuda.synchronize()
    assert (y.cpu().t() - ref).abs().max().item() < 5e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.float32, 2e-5)])
def test_decode_attention_all_cache_lengths(dtype, tol):
    """the AR decode-step attention kernel alone (reference t2s_model.py:176-221: softmax(q.K^T/sqrt(32)).V over the cached
    keys) vs torch fp32 on the same (dtype-rounded) inputs, for cache lengths on every side of the kernel's phase
    boundaries (speculative groups, second batch, pipelined tail: 1 key ... the reference's 1500-token limit), ragged per
    row, one inactive row left untouched.  max-abs tolerance in the table (|out| <= 1)."""
    import ctypes as C
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(5)
    H, HD, smax = 4, 32, 1536
    lens = [0, 1, 15, 16, 63, 64, 191, 192, 255, 256, 257, 511, 512, 513, 640, 767, 1023, 1024, 1499, 1534]
    B = len(lens) + 1
    q = torch.randn(B, H * HD, device=DEV).to(dtype)
    kc = torch.randn(B, H, smax, HD, device=DEV).to(dtype)
    vc = torch.randn(B, H, smax, HD, device=DEV).to(dtype)
    kv_len = torch.tensor(lens + [100], dtype=torch.int32, device=DEV)
    active = torch.ones(B, dtype=torch.int32, device=DEV)
    active[-1] = 0
    out = torch.full((B, H * HD), 7.0, device=DEV, dtype=dtype)
    code = _lib.dtype_code(dtype)
    _lib.check(_lib.lib().gsv_op_decode_attn(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), kv_len.data_ptr(), active.data_ptr(),
                                             B, H, smax, code, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.all(out[-1] == 7.0)
    qf = q.float().view(B, H, 1, HD)
    for b, n in enumerate(lens):
        k = kc[b, :, : n + 1].float()
        v = vc[b, :, : n + 1].float()
        p = torch.softmax((qf[b] @ k.transpose(1, 2)) / HD ** 0.5, dim=-1)
        ref = (p @ v).reshape(-1)
        err = (out[b].float() - ref).abs().max().item()
        assert err <= tol, (n, err)


@pytest.mark.parametrize("C_,k,dil,T,accum", [(16, 11, 5, 3000, False), (16, 3, 1, 256, True), (16, 7, 3, 1025, False), (16, 11, 1, 4500, True),
                                              (32, 11, 5, 2111, True), (32, 11, 5, 5000, True), (32, 7, 1, 777, False), (32, 3, 5, 40000, False)])
def test_conv_pair_matches_two_launches(C_, k, dil, T, accum):
    """Fused ResBlock pair (conv_pair.hip) vs the two-launch path it replaces -- convs1 (lrelu in, dilation d) then convs2 (lrelu in,
    + x, * scale, optional accumulate) through gsv_op_conv1d: bit-identical fp16 (same rounding points, same MFMA order; one-ulp exceptions with accumulate, below);
    and vs torch fp32 of the reference's ResBlock1 pair within fp16 tolerance.  Tile edges, zero padding of the intermediate,
    ragged last tile and the persistent multi-tile loop (40000 rows) are all covered."""
    from gsv import _lib
    _lib.init(0)
    torch.manual_seed(C_ * 100 + k * 10 + dil)
    x = torch.randn(C_, T)
    w1, w2 = torch.randn(C_, C_, k) / (C_ * k) ** 0.5, torch.randn(C_, C_, k) / (C_ * k) ** 0.5
    b1, b2 = torch.randn(C_) * 0.1, torch.randn(C_) * 0.1
    y0 = torch.randn(C_, T) if accum else None
    scale = 1.0 / 3.0 if accum else 1.0
    # two launches (what the generator did before)
    xh = x.half().float()
    t = _conv(xh, w1, b1, torch.float16, dil=dil, pre_lrelu=0.1)
    two = _conv(t, w2, b2, torch.float16, dil=1, pre_lrelu=0.1, res=xh, scale=scale, accumulate=y0.half().float() if accum else None)
    # fused
    xd = x.t().contiguous().to(DEV, torch.float16)
    pk = lambda w: w.permute(0, 2, 1).reshape(C_, k * C_).contiguous().to(DEV, torch.float16)
    w1d, w2d, b1d, b2d = pk(w1), pk(w2), b1.to(DEV), b2.to(DEV)
    yd = y0.t().contiguous().to(DEV, torch.float16) if accum else torch.zeros(T, C_, device=DEV, dtype=torch.float16)
    _lib.check(_lib.lib().gsv_op_conv_pair(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), yd.data_ptr(),
                                           T, C_, k, dil, scale, 1 if accum else 0, None), "gsv_op_conv_pair")
    torch.cuda.synchronize()
    fused = yd.float().cpu().t()
    if accum:
        # with scale = 1/3 and an accumulate operand the compilers' contraction of (acc + b + x) * scale + y differs between the
        # two kernels: a handful of results differ by one fp16 ulp (measured 12 of 72 000); everything else is identical
        d = (fused - two).abs()
        assert (d <= 1e-3 * (1 + two.abs())).all() and (d > 0).float().mean() < 1e-3
    else:
        assert torch.equal(fused, two), f"max diff {(fused - two).abs().max()}"
    ref = F.conv1d(F.leaky_relu(xh, 0.1).unsqueeze(0), w1.half().float(), b1, dilation=dil, padding=(k - 1) // 2 * dil)
    ref = F.conv1d(F.leaky_relu(ref, 0.1), w2.half().float(), b2, padding=(k - 1) // 2)[0]
    ref = (ref + xh) * scale + (y0.half().float() if accum else 0)
    assert (fused - ref).abs().max() <= 2e-2
    with pytest.raises(RuntimeError):
        _lib.check(_lib.lib().gsv_op_conv_pair(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), yd.data_ptr(),
                                               T, 48, k, dil, scale, 0, None), "gsv_op_conv_pair")


@pytest.mark.parametrize("k,dil,mode", [(3, 1, "plain"), (7, 3, "res"), (11, 5, "res"), (11, 1, "res_acc"), (7, 5, "acc")])
def test_wide_persistent_conv_matches_tile_kernel(k, dil, mode):
    """conv_wide_f16_kernel (persistent, C = 128, T >= 16384) vs conv_lds_kernel (one tile per workgroup) on the same data:
    the tile kernel is reached by convolving overlapping pieces shorter than 16384 steps, whose interior rows must be
    bit-identical to the persistent kernel's (one-ulp exceptions with accumulate); and vs torch fp32 within fp16 tolerance.  T = 20000 is ragged (last tile 32 rows)."""
    torch.manual_seed(k * 7 + dil)
    C_, T = 128, 20000
    x = torch.randn(C_, T).half().float()
    w = (torch.randn(C_, C_, k) / (C_ * k) ** 0.5).half().float()
    b = torch.randn(C_) * 0.1
    res = torch.randn(C_, T).half().float() if "res" in mode else None
    y0 = torch.randn(C_, T).half().float() if "acc" in mode else None
    scale = 1.0 / 3.0 if "acc" in mode else 1.0
    full = _conv(x, w, b, torch.float16, dil=dil, pre_lrelu=0.1, res=res, scale=scale, accumulate=y0)
    halo = (k - 1) // 2 * dil
    for lo, hi in ((0, 12000), (8000, 20000)):
        piece = _conv(x[:, lo:hi], w, b, torch.float16, dil=dil, pre_lrelu=0.1, res=res[:, lo:hi] if res is not None else None,
                      scale=scale, accumulate=y0[:, lo:hi] if y0 is not None else None)
        a, z = (0 if lo == 0 else halo), (hi - lo if hi == T else hi - lo - halo)     # rows whose receptive field lies inside the piece
        pa, fu = piece[:, a:z], full[:, lo + a:lo + z]
        if "acc" in mode:     # scale 1/3 + accumulate: the two epilogues' FMA contraction differs, one fp16 ulp on a few elements
            d = (pa - fu).abs()
            assert (d <= 1e-3 * (1 + fu.abs())).all() and (d > 0).float().mean() < 1e-3
        else:
            assert torch.equal(pa, fu), f"max diff {(pa - fu).abs().max()}"
    ref = F.conv1d(F.leaky_relu(x, 0.1).unsqueeze(0), w, b, dilation=dil, padding=halo)[0]
    if res is not None:
        ref = ref + res
    ref = ref * scale + (y0 if y0 is not None else 0)
    assert (full - ref).abs().max() <= 2e-2
