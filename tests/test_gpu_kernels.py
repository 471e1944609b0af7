"""Per-kernel parity through the C-ABI (ctypes) against torch-CPU fp32 / the oracle.

Integer and index outputs must be bit-exact.  fp32-mode floats: 1e-3 relative (task tolerance; the
observed error is ~1e-6).  bf16-mode floats: inputs are rounded to bf16 for both sides, so only the
fp32-accumulation order differs: 2e-2 relative on outputs of O(1).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    import image_captioning_amd as m
    return m._lib


@pytest.fixture(scope="module")
def W():
    import image_captioning_amd as m
    return m.weights


def _tol(dtype):
    return dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def _conv(L, W, x_nchw, w, bias, stride, pad, relu, dtype, res_nchw=None):
    """Run mic_conv2d on an NCHW problem and return NCHW fp32."""
    B, Ci, H, Wd = x_nchw.shape
    Co, _, kh, kw = w.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wm = W.conv_matrix(w, dtype, DEV)
    out = torch.empty(B, OH, OW, Co, dtype=dtype, device=DEV)
    res = None if res_nchw is None else res_nchw.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    b = None if bias is None else bias.to(DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.res, d.out = x.data_ptr(), wm.data_ptr(), L.ptr(b), L.ptr(res), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.dtype_code(dtype)
    d.M, d.N, d.K = B * OH * OW, Co, kh * kw * Ci
    d.OH, d.OW, d.IH, d.IW = OH, OW, H, Wd
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = H * Wd * Ci, Wd * Ci, Ci
    d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = kw, Ci, stride, pad, pad
    d.ldc, d.ld_res, d.relu = Co, Co, int(relu)
    L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()), "mic_conv2d")
    return out.float().cpu().permute(0, 3, 1, 2)


CONV_CASES = [
    # B, Ci, H, W, Co, k, stride, pad, relu, residual
    (2, 64, 14, 14, 64, 1, 1, 0, True, False),      # 1x1
    (2, 64, 14, 14, 256, 1, 1, 0, True, True),      # 1x1 + residual + relu (bottleneck conv3)
    (2, 64, 15, 13, 96, 3, 1, 1, True, False),      # 3x3 pad 1, odd sizes, Co not a tile multiple
    (3, 128, 14, 14, 128, 3, 2, 1, True, False),    # 3x3 stride 2
    (2, 256, 14, 14, 512, 1, 2, 0, False, False),   # strided 1x1 shortcut, no activation
    (1, 32, 7, 7, 40, 3, 1, 1, False, False),       # tiny: one partial tile in M and N
    (9, 128, 28, 28, 128, 3, 1, 1, True, False),    # M = 7056: many 128x128 tiles with an M tail
    (40, 64, 28, 28, 256, 1, 1, 0, True, True),     # 128x128 tile path with residual
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(L, W, dtype, case):
    B, Ci, H, Wd, Co, k, s, p, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Ci, H, Wd, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, b, s, p)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        if dtype == torch.bfloat16:
            res = res.bfloat16().float()
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    got = _conv(L, W, x, w, b, s, p, relu, dtype, res)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [
    # B, main channels, shortcut channels, H (shortcut input), stride, Co
    (2, 64, 64, 14, 1, 256),     # short K, N a multiple of 256
    (3, 128, 256, 15, 2, 512),   # stride-2 shortcut over an odd map, K = 384
    (2, 64, 32, 9, 2, 96),       # N not a multiple of 256 (128x128 tiles), partial tiles
    (5, 256, 512, 14, 2, 1024),  # K = 768
])
def test_conv2d_second_k_segment(L, W, dtype, case):
    """out = relu(conv1x1(y) + conv1x1_stride_s(x) + bias): the fused tail of a bottleneck's first block."""
    B, C1, C2, H, st, Co = case
    g = torch.Generator().manual_seed(sum(case))
    OH = (H - 1) // st + 1
    y = torch.randn(B, C1, OH, OH, generator=g)
    x = torch.randn(B, C2, H, H, generator=g)
    w1 = torch.randn(Co, C1, 1, 1, generator=g) / math.sqrt(C1)
    w2 = torch.randn(Co, C2, 1, 1, generator=g) / math.sqrt(C2)
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:
        y, x, w1, w2 = (t.bfloat16().float() for t in (y, x, w1, w2))
    ref = F.relu(F.conv2d(y, w1, b) + F.conv2d(x, w2, None, st))
    yd = y.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wm = W.matrix(torch.cat([w1.reshape(Co, C1), w2.reshape(Co, C2)], dim=1), dtype, DEV)
    bd = b.to(DEV)
    out = torch.empty(B, OH, OH, Co, dtype=dtype, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out = yd.data_ptr(), wm.data_ptr(), bd.data_ptr(), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.dtype_code(dtype)
    d.M, d.N, d.K = B * OH * OH, Co, C1
    d.OH, d.OW, d.IH, d.IW = OH, OH, OH, OH
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = OH * OH * C1, OH * C1, C1
    d.kw_taps, d.run, d.stride = 1, C1, 1
    d.ldc, d.relu = Co, 1
    d.in2, d.K2, d.stride2 = xd.data_ptr(), C2, st
    d.in2_img_stride, d.in2_row_stride, d.in2_pix_stride = H * H * C2, H * C2, C2
    L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()), "mic_conv2d")
    np.testing.assert_allclose(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), **_tol(dtype))
    d.K2 = 24  # not a multiple of the k-tile
    assert L.lib.mic_conv2d(C.byref(d), L.stream_ptr()) == -1


def test_conv2d_rejects_bad_arguments(L):
    d = L.ConvDesc()
    assert L.lib.mic_conv2d(C.byref(d), None) == -1
    assert b"null" in L.lib.mic_last_error()
    x = torch.zeros(64, 48, device=DEV)
    d.in_ = d.w = d.out = x.data_ptr()
    d.dtype = d.out_dtype = 0
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = 64, 48, 40, 1, 1, 1, 1
    d.run, d.kw_taps, d.stride, d.ldc, d.in_pix_stride, d.in_img_stride = 40, 1, 1, 48, 4, 48
    assert L.lib.mic_conv2d(C.byref(d), None) == -1  # K=40 is not a multiple of the 16-element fp32 k-tile
    with pytest.raises(L.MicError):
        L.check(-1, "x")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 200, 64), (3, 3500, 512), (768, 512, 1024), (130, 1536, 512)])
def test_linear(L, W, dtype, M, N, K):
    """The 1x1 / single-pixel case used by every Linear on the decode path, fp32 output."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xd, wd, bd, rd = x.to(DEV, dtype), W.matrix(w, dtype, DEV), b.to(DEV), r.to(DEV)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.res, d.out = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), out.data_ptr()
    d.dtype, d.out_dtype, d.res_dtype = L.dtype_code(dtype), L.F32, L.F32
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, N, K, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc, d.ld_res = K, 4, 1, K, 1, N, N
    L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()))
    ref = x @ w.t() + b + r
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_maxpool_avgpool(L, dtype):
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 37, 41, generator=g)
    imgd = img.to(DEV)
    out = torch.full((2, 37 + 6, 41 + 9, 4), 7.0, dtype=dtype, device=DEV)
    L.check(L.lib.mic_pack_image(imgd.data_ptr(), out.data_ptr(), L.dtype_code(dtype), 2, 37, 41, 3, 3, 3, 6, L.stream_ptr()))
    ref = torch.zeros(2, 43, 50, 4)
    ref[:, 3:40, 3:44, :3] = img.permute(0, 2, 3, 1)
    np.testing.assert_array_equal(out.float().cpu().numpy(), ref.to(dtype).float().numpy())

    x = torch.randn(2, 64, 17, 19, generator=g).to(dtype)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    po = torch.empty(2, 9, 10, 64, dtype=dtype, device=DEV)
    L.check(L.lib.mic_maxpool3x3s2(xd.data_ptr(), po.data_ptr(), L.dtype_code(dtype), 2, 17, 19, 64, L.stream_ptr()))
    np.testing.assert_array_equal(po.float().cpu().permute(0, 3, 1, 2).numpy(), F.max_pool2d(x.float(), 3, 2, 1).numpy())

    ao = torch.empty(2, 64, dtype=dtype, device=DEV)
    L.check(L.lib.mic_global_avgpool(xd.data_ptr(), ao.data_ptr(), L.dtype_code(dtype), 2, 17 * 19, 64, 64, L.stream_ptr()))
    np.testing.assert_allclose(ao.float().cpu().numpy(), x.float().mean(dim=(2, 3)).numpy(), **_tol(dtype))


def test_embed_layernorm_lstm_cell_gather(L):
    g = torch.Generator().manual_seed(4)
    V, H, N, T = 50, 64, 5, 7
    tok, pos = torch.randn(V, H, generator=g), torch.randn(T, H, generator=g)
    ids = torch.randint(0, V, (N, T), generator=g, dtype=torch.int32)
    x = torch.empty(N, H, device=DEV)
    xt = torch.empty(N, H, dtype=torch.bfloat16, device=DEV)
    tokd, posd, idsd = tok.to(DEV), pos.to(DEV), ids.to(DEV)
    L.check(L.lib.mic_embed_pos(idsd.data_ptr(), T, 3, tokd.data_ptr(), posd.data_ptr(), x.data_ptr(), xt.data_ptr(), H, L.BF16, N, H, V,
                                L.stream_ptr()))
    ref = tok[ids[:, 3].long()] + pos[3]
    np.testing.assert_array_equal(x.cpu().numpy(), ref.numpy())
    np.testing.assert_array_equal(xt.float().cpu().numpy(), ref.bfloat16().float().numpy())

    y = torch.randn(N, H, generator=g) * 3 + 1
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    yd, o, gd_, bd_ = y.to(DEV), torch.empty(N, H, device=DEV), gam.to(DEV), bet.to(DEV)
    L.check(L.lib.mic_layernorm(yd.data_ptr(), gd_.data_ptr(), bd_.data_ptr(), o.data_ptr(), None, 0, L.F32, N, H, L.stream_ptr()))
    np.testing.assert_allclose(o.cpu().numpy(), F.layer_norm(y, (H,), gam, bet).numpy(), rtol=1e-5, atol=1e-5)

    Hd = 24
    gates, c0 = torch.randn(N, 4 * Hd, generator=g), torch.randn(N, Hd, generator=g)
    cd, hd_, gd = c0.to(DEV), torch.zeros(N, Hd, device=DEV), gates.to(DEV)
    ht = torch.zeros(N, 40, device=DEV)
    L.check(L.lib.mic_lstm_cell(gd.data_ptr(), cd.data_ptr(), hd_.data_ptr(), ht.data_ptr(), 40, None, 0, L.F32, N, Hd, L.stream_ptr()))
    i_, f_, g_, o_ = gates.chunk(4, dim=1)
    cr = torch.sigmoid(f_) * c0 + torch.sigmoid(i_) * torch.tanh(g_)
    hr = torch.sigmoid(o_) * torch.tanh(cr)
    np.testing.assert_allclose(cd.cpu().numpy(), cr.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(hd_.cpu().numpy(), hr.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(ht[:, :Hd].cpu().numpy(), hd_.cpu().numpy())

    src = torch.randn(6, 10, generator=g).to(DEV)
    idx = torch.tensor([5, 0, 0, 3, 2, 2], dtype=torch.int32, device=DEV)
    dst = torch.empty_like(src)
    L.check(L.lib.mic_gather_rows(src.data_ptr(), dst.data_ptr(), idx.data_ptr(), 6, 40, L.stream_ptr()))
    np.testing.assert_array_equal(dst.cpu().numpy(), src.cpu().numpy()[idx.cpu().numpy()])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,heads,Tc", [(64, 2, 9), (512, 2, 9), (256, 8, 9), (512, 1, 9), (512, 2, 32), (512, 8, 33), (256, 4, 32)])
def test_attention_steps(L, dtype, H, heads, Tc):
    """Self form against a cache filled step by step with a random ancestry; cross form against
    per-image K/V; both versus softmax(q k^T / sqrt(hd)) v computed on the CPU."""
    g = torch.Generator().manual_seed(H + heads)
    N, S, rpi = 6, 49, 3
    hd = H // heads
    code = L.dtype_code(dtype)

    def rnd(*s):
        t = torch.randn(*s, generator=g)
        return t.to(dtype).float() if dtype == torch.bfloat16 else t

    kc = torch.zeros(N, Tc, H, dtype=dtype, device=DEV)
    vc = torch.zeros(N, Tc, H, dtype=dtype, device=DEV)
    o = torch.empty(N, H, dtype=dtype, device=DEV)
    hist_k, hist_v = torch.zeros(Tc, N, H), torch.zeros(Tc, N, H)  # what slot n wrote at step t
    anc = torch.zeros(N, Tc, dtype=torch.int32)
    for t in range(Tc):
        qkv = torch.randn(N, 3 * H, generator=g)
        hist_k[t], hist_v[t] = qkv[:, H:2 * H].to(dtype).float(), qkv[:, 2 * H:].to(dtype).float()
        if t > 0:  # random reorder: row n continues the lineage of a random parent
            par = torch.randint(0, N, (N,), generator=g)
            anc[:, :t] = anc[par, :t]
            anc[:, t - 1] = par.to(torch.int32)
        qd, ad = qkv.to(DEV), anc.to(DEV)
        L.check(L.lib.mic_self_attn_step(qd.data_ptr(), kc.data_ptr(), vc.data_ptr(), ad.data_ptr(), t, Tc, o.data_ptr(), H, code, N, H,
                                         heads, L.stream_ptr()))
        ref = torch.zeros(N, H)
        for n in range(N):
            ks = torch.stack([hist_k[p, anc[n, p]] if p < t else hist_k[t, n] for p in range(t + 1)])
            vs = torch.stack([hist_v[p, anc[n, p]] if p < t else hist_v[t, n] for p in range(t + 1)])
            for h in range(heads):
                sl = slice(h * hd, (h + 1) * hd)
                w = torch.softmax((qkv[n, sl] * math.sqrt(1.0 / hd)) @ ks[:, sl].t(), dim=-1)
                ref[n, sl] = w @ vs[:, sl]
        np.testing.assert_allclose(o.float().cpu().numpy(), ref.numpy(), **_tol(dtype))

    kv = rnd(N // rpi, S, 2 * H)
    q = torch.randn(N, H, generator=g)
    attn = torch.full((N, S), 0.25, device=DEV)
    qd, kvd = q.to(DEV), kv.to(DEV, dtype)  # named: a temporary's block could be reused before the launch
    L.check(L.lib.mic_cross_attn_step(qd.data_ptr(), H, kvd.data_ptr(), S, rpi, o.data_ptr(), H, attn.data_ptr(), S,
                                      0.5, 1, code, N, H, heads, L.stream_ptr()))
    ref, wref = torch.zeros(N, H), torch.zeros(N, S)
    for n in range(N):
        k_, v_ = kv[n // rpi, :, :H], kv[n // rpi, :, H:]
        for h in range(heads):
            sl = slice(h * hd, (h + 1) * hd)
            w = torch.softmax((q[n, sl] * math.sqrt(1.0 / hd)) @ k_[:, sl].t(), dim=-1)
            ref[n, sl] = w @ v_[:, sl]
            wref[n] += w / heads
    np.testing.assert_allclose(o.float().cpu().numpy(), ref.numpy(), **_tol(dtype))
    np.testing.assert_allclose(attn.cpu().numpy(), 0.25 + 0.5 * wref.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype,H,heads", [(torch.bfloat16, 512, 2), (torch.bfloat16, 512, 8), (torch.float32, 256, 4)])
@pytest.mark.parametrize("rpi,S", [(2, 49), (3, 64), (4, 17), (5, 49), (7, 49), (8, 33)])
def test_cross_attention_one_workgroup_per_image(L, dtype, H, heads, rpi, S):
    """Shapes that take the 16-wave per-image kernel (hidden size = 64 sixteen-byte chunks, S <= 64): every
    beam count, a full and a ragged last key group, with and without the attention-weight output."""
    g = torch.Generator().manual_seed(H + heads + rpi + S)
    B, hd, code = 5, H // heads, L.dtype_code(dtype)
    N = B * rpi
    kv = torch.randn(B, S, 2 * H, generator=g)
    if dtype == torch.bfloat16:
        kv = kv.bfloat16().float()
    q = torch.randn(N, H, generator=g)
    qd, kvd = q.to(DEV), kv.to(DEV, dtype)
    o = torch.empty(N, H, dtype=dtype, device=DEV)
    attn = torch.full((N, S), 7.0, device=DEV)
    L.check(L.lib.mic_cross_attn_step(qd.data_ptr(), H, kvd.data_ptr(), S, rpi, o.data_ptr(), H, attn.data_ptr(), S,
                                      1.0, 0, code, N, H, heads, L.stream_ptr()))
    qh = (q * math.sqrt(1.0 / hd)).view(B, rpi, heads, hd)
    k_, v_ = kv[:, :, :H].view(B, S, heads, hd), kv[:, :, H:].view(B, S, heads, hd)
    w = torch.softmax(torch.einsum("brhd,bshd->brhs", qh, k_), dim=-1)
    ref = torch.einsum("brhs,bshd->brhd", w, v_).reshape(N, H)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref.numpy(), **_tol(dtype))
    np.testing.assert_allclose(attn.cpu().numpy(), w.mean(dim=2).reshape(N, S).numpy(), rtol=1e-4, atol=1e-6)
    o2 = torch.empty(N, H, dtype=dtype, device=DEV)
    L.check(L.lib.mic_cross_attn_step(qd.data_ptr(), H, kvd.data_ptr(), S, rpi, o2.data_ptr(), H, None, 0,
                                      1.0, 0, code, N, H, heads, L.stream_ptr()))
    assert torch.equal(o, o2)


@pytest.mark.parametrize("dtype,H,heads", [(torch.bfloat16, 512, 2), (torch.bfloat16, 512, 1), (torch.float32, 256, 2)])
@pytest.mark.parametrize("rpi,S", [(1, 49), (3, 49), (4, 64), (5, 33), (8, 49)])
def test_cross_attention_folded_sublayer(L, W, dtype, H, heads, rpi, S):
    """y = LN(r) + MHA(LN(r), kv, kv) with the q / out projections folded into per-image K', V' (built here the way
    mic_tr_cross_kv does: per-head GEMMs + the bias kernel), against the unfolded mathematics on the CPU."""
    g = torch.Generator().manual_seed(H + heads + rpi + S)
    B, hd, code = 4, H // heads, L.dtype_code(dtype)
    N, parts = B * rpi, (H + 127) // 128

    def q_(t):
        return t.to(dtype).float() if dtype == torch.bfloat16 else t

    r = torch.randn(N, H, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    wq, bq = torch.randn(H, H, generator=g) / math.sqrt(H), 0.1 * torch.randn(H, generator=g)
    wo, bo = torch.randn(H, H, generator=g) / math.sqrt(H), 0.1 * torch.randn(H, generator=g)
    kv = q_(torch.randn(B, S, 2 * H, generator=g))
    sc = 1.0 / math.sqrt(hd)
    # device-side preparation through the library: per-head folded weights -> K', V' (GEMM), c (bias kernel)
    kvd = kv.to(DEV, dtype)
    kf = torch.empty(B, S, heads, H, dtype=dtype, device=DEV)
    vf = torch.empty(B, S, heads, H, dtype=dtype, device=DEV)
    cc = torch.empty(B, S, heads, device=DEV)
    keep = []
    sz = kvd.element_size()
    for h in range(heads):
        for dst, wmat, off in ((kf, (wq[h * hd:(h + 1) * hd, :].T * sc).contiguous(), h * hd), (vf, wo[:, h * hd:(h + 1) * hd].contiguous(), H + h * hd)):
            wm = W.matrix(wmat, dtype, DEV)
            keep.append(wm)
            d = L.ConvDesc()
            d.in_, d.w, d.out = kvd.data_ptr() + off * sz, wm.data_ptr(), dst.data_ptr() + h * H * sz
            d.dtype = d.out_dtype = d.res_dtype = code
            d.M, d.N, d.K = B * S, H, hd
            d.OH = d.OW = d.IH = d.IW = 1
            d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc = 2 * H, 4, 1, hd, 1, heads * H
            L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()))
    qbs = (bq * sc).to(DEV)
    L.check(L.lib.mic_cross_fold_bias(kvd.data_ptr(), qbs.data_ptr(), cc.data_ptr(), code, B * S, H, heads, L.stream_ptr()))
    # statistics of r per 128 columns, as a producer launch would have written them
    rp = r.view(N, parts, -1)
    stats = torch.stack([rp.mean(-1), ((rp - rp.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).contiguous()
    rd, sd_, gd, bd, bod = r.to(DEV), stats.to(DEV), gamma.to(DEV), beta.to(DEV), bo.to(DEV)
    y = torch.empty(N, H, device=DEV)
    so = torch.empty(N, parts, 2, device=DEV)
    attn = torch.zeros(N, S, device=DEV)
    xv = L.LnRef(rd.data_ptr(), sd_.data_ptr(), gd.data_ptr(), bd.data_ptr(), H, parts, H)
    L.check(L.lib.mic_cross_attn_folded(C.byref(xv), kf.data_ptr(), vf.data_ptr(), cc.data_ptr(), bod.data_ptr(), S, rpi, y.data_ptr(),
                                        so.data_ptr(), attn.data_ptr(), S, 1.0, 0, code, N, H, heads, L.stream_ptr()))
    x = F.layer_norm(r, (H,), gamma, beta, 1e-5)
    qh = (F.linear(x, wq, bq) * sc).view(B, rpi, heads, hd)
    k_, v_ = kv[:, :, :H].view(B, S, heads, hd), kv[:, :, H:].view(B, S, heads, hd)
    w = torch.softmax(torch.einsum("brhd,bshd->brhs", qh, k_), dim=-1)
    ref = x + F.linear(torch.einsum("brhs,bshd->brhd", w, v_).reshape(N, H), wo, bo)
    tol = dict(rtol=1e-3, atol=2e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), **tol)
    np.testing.assert_allclose(attn.cpu().numpy(), w.mean(dim=2).reshape(N, S).numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=1e-4)
    yp = y.cpu().view(N, parts, -1)
    np.testing.assert_allclose(so.cpu()[..., 0].numpy(), yp.mean(-1).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(so.cpu()[..., 1].numpy(), ((yp - yp.mean(-1, keepdim=True)) ** 2).sum(-1).numpy(), rtol=1e-4, atol=1e-4)
    # in place (y aliasing the raw stream), as the step driver calls it
    L.check(L.lib.mic_cross_attn_folded(C.byref(xv), kf.data_ptr(), vf.data_ptr(), cc.data_ptr(), bod.data_ptr(), S, rpi, rd.data_ptr(),
                                        so.data_ptr(), None, 0, 1.0, 0, code, N, H, heads, L.stream_ptr()))
    assert torch.equal(rd, y)
    assert L.lib.mic_cross_attn_folded_supported(code, H, 4, S, rpi) == 0 and L.lib.mic_cross_attn_folded_supported(code, H + 64, heads, S, rpi) == 0


@pytest.mark.parametrize("V", [200, 3500, 4096, 4097, 6501])
@pytest.mark.parametrize("K", [1, 3, 5, 8])
def test_logsoftmax_topk(L, V, K):
    g = torch.Generator().manual_seed(V + K)
    N = 37
    x = torch.randn(N, V, generator=g) * 3
    x[:, [0, 1, 3]] = -1e9          # banned ids as the reference biases them
    x[5, 10] = x[5, 20] = x[5, 30] = 50.0  # exact ties at the top: lower index first
    x[6] = 0.0                      # a constant row
    xd = x.to(DEV)
    val = torch.empty(N, K, device=DEV)
    idx = torch.empty(N, K, dtype=torch.int32, device=DEV)
    lse = torch.empty(N, device=DEV)
    L.check(L.lib.mic_logsoftmax_topk(xd.data_ptr(), V, N, V, K, 1.0, 0, val.data_ptr(), idx.data_ptr(), lse.data_ptr(), L.stream_ptr()))
    lp = F.log_softmax(x, dim=-1)
    # reference order: value descending, index ascending on ties (stable sort of the negated values)
    order = torch.argsort(-lp.double(), dim=-1, stable=True)[:, :K]
    got_idx = idx.cpu().long()
    # indices must agree wherever the reference values are distinct; tie rows checked explicitly
    np.testing.assert_allclose(val.cpu().numpy(), lp.gather(1, got_idx).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(val.cpu().numpy(), lp.gather(1, order).numpy(), rtol=1e-5, atol=1e-5)
    assert got_idx[5, : min(K, 3)].tolist() == [10, 20, 30][: min(K, 3)]
    assert got_idx[6].tolist() == list(range(K))
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(x, dim=-1).numpy(), rtol=1e-5)
    # raw ranking (LSTM greedy): same winners as argmax of the logits
    val1 = torch.empty(N, device=DEV)
    idx1 = torch.empty(N, dtype=torch.int32, device=DEV)
    L.check(L.lib.mic_logsoftmax_topk(xd.data_ptr(), V, N, V, 1, 1.0, 1, val1.data_ptr(), idx1.data_ptr(), None, L.stream_ptr()))
    assert idx1.cpu().long().tolist() == x.argmax(dim=-1).tolist()


def test_sample_multinomial(L):
    from oracle import captioner_oracle as O
    g = torch.Generator().manual_seed(9)
    N, V = 64, 3500
    x = torch.randn(N, V, generator=g) * 2
    u = torch.rand(N, generator=g)
    u[0], u[1] = 0.0, 0.99999994
    xd, ud = x.to(DEV), u.to(DEV)
    val = torch.empty(N, device=DEV)
    idx = torch.empty(N, dtype=torch.int32, device=DEV)
    L.check(L.lib.mic_sample_multinomial(xd.data_ptr(), V, N, V, 1 / 0.7, ud.data_ptr(), val.data_ptr(), idx.data_ptr(), L.stream_ptr()))
    lp = F.log_softmax(x / 0.7, dim=-1)
    ref = O.inverse_cdf_sampler(u[None].double())(lp.exp(), 0).squeeze(1)
    assert idx.cpu().long().tolist() == ref.tolist()
    np.testing.assert_allclose(val.cpu().numpy(), lp.gather(1, ref[:, None]).squeeze(1).numpy(), rtol=1e-5, atol=1e-5)


def _py_beam_select(score, length, seqs, top_val, top_idx, K, n_live_slots, eos, in_place):
    """The reference's candidate pooling and stable sort for one image (models/transformer.py:613-628,
    models/image_captioner.py:128-149), on explicit top-k inputs."""
    beams = [(score[i], seqs[i][: length[i]], i) for i in range(n_live_slots)]
    grown, fin = [], []
    pool = []
    for sc, sq, i in beams:
        if sq[-1] == eos:
            (pool if in_place else fin).append((sc, sq, i))
            continue
        for j in range(K):
            (pool if in_place else grown).append((sc + float(top_val[i][j]), sq + [int(top_idx[i][j])], i))
    if not in_place:
        pool = grown + fin
    pool.sort(key=lambda b: b[0] / (len(b[1]) ** 0.5), reverse=True)
    return pool[:K]


@pytest.mark.parametrize("in_place", [0, 1])
@pytest.mark.parametrize("K", [2, 3, 5, 8])
def test_beam_select_bit_exact(L, K, in_place):
    """Crafted pools: duplicated fp32 log-probs give exact fp64 ties between candidates of different
    beams and between finished and live beams; the survivors, their order, scores and parents must
    equal the Python sort."""
    rng = np.random.default_rng(K * 10 + in_place)
    B, t_cap, t, eos = 23, 12, 4, 2
    N = B * K
    vals = np.array([-0.25, -0.5, -0.75, -1.0, -1.5, -2.0, -3.0], dtype=np.float32)
    score = rng.choice(vals, size=N).astype(np.float64) * 2
    length = np.full(N, t + 1, dtype=np.int32)
    seq = rng.integers(4, 50, size=(N, t_cap)).astype(np.int32)
    for n in range(N):  # a third of the slots already finished, at shorter lengths
        if rng.random() < 0.35:
            length[n] = rng.integers(2, t + 2)
            seq[n, length[n] - 1] = eos
    top_val = np.sort(rng.choice(vals, size=(N, K)), axis=1)[:, ::-1].copy()
    top_idx = rng.integers(2, 50, size=(N, K)).astype(np.int32)  # includes EOS now and then
    anc = rng.integers(0, N, size=(N, t_cap)).astype(np.int32)
    sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64, device=DEV)

    def dev(a):
        return torch.from_numpy(a).to(DEV)

    cur_t = [dev(score), dev(length), dev(seq), dev(anc)]
    nxt_t = [torch.zeros_like(cur_t[0]), torch.zeros_like(cur_t[1]), torch.zeros_like(cur_t[2]), torch.zeros_like(cur_t[3])]
    cur, nxt = L.BeamState(*[x.data_ptr() for x in cur_t]), L.BeamState(*[x.data_ptr() for x in nxt_t])
    if in_place:
        cur.anc = nxt.anc = None
    parent = torch.zeros(N, dtype=torch.int32, device=DEV)
    live = torch.zeros(1, dtype=torch.int32, device=DEV)
    tvd, tid = dev(top_val), dev(top_idx)
    L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tvd.data_ptr(), tid.data_ptr(), parent.data_ptr(),
                                  live.data_ptr(), sqrt_tab.data_ptr(), B, K, K, t, t_cap, eos, in_place, L.stream_ptr()))
    g_score, g_len, g_seq, g_anc = (x.cpu().numpy() for x in nxt_t)
    g_par = parent.cpu().numpy()
    n_live = 0
    for b in range(B):
        sl = slice(b * K, (b + 1) * K)
        ref = _py_beam_select(score[sl].tolist(), length[sl].tolist(), seq[sl].tolist(), top_val[sl], top_idx[sl], K, K, eos, in_place)
        alive = False
        for r, (sc, sq, i) in enumerate(ref):
            n = b * K + r
            assert g_score[n] == sc and g_len[n] == len(sq) and g_seq[n, : len(sq)].tolist() == sq and g_par[n] == b * K + i
            grew = len(sq) == length[b * K + i] + 1
            alive |= grew and sq[-1] != eos
            if not in_place:
                if grew:
                    assert g_anc[n, :t].tolist() == anc[b * K + i, :t].tolist() and g_anc[n, t] == b * K + i
                else:
                    assert g_anc[n, : len(sq) - 1].tolist() == anc[b * K + i, : len(sq) - 1].tolist()
        n_live += alive
    assert int(live.item()) == n_live


def test_beam_select_first_step_uses_one_slot(L):
    K, B, t_cap = 3, 2, 6
    N = B * K
    st = [torch.zeros(N, dtype=torch.float64, device=DEV), torch.ones(N, dtype=torch.int32, device=DEV),
          torch.ones(N, t_cap, dtype=torch.int32, device=DEV), torch.zeros(N, t_cap, dtype=torch.int32, device=DEV)]
    nx = [torch.zeros_like(x) for x in st]
    cur, nxt = L.BeamState(*[x.data_ptr() for x in st]), L.BeamState(*[x.data_ptr() for x in nx])
    tv = torch.tensor([[-1.0, -2.0, -3.0]] * N, device=DEV)
    ti = torch.tensor([[7, 8, 9]] * N, dtype=torch.int32, device=DEV)
    parent = torch.zeros(N, dtype=torch.int32, device=DEV)
    sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64, device=DEV)
    L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tv.data_ptr(), ti.data_ptr(), parent.data_ptr(), None, sqrt_tab.data_ptr(),
                                  B, K, 1, 0, t_cap, 2, 0, L.stream_ptr()))
    assert nx[2][:, :2].cpu().tolist() == [[1, 7], [1, 8], [1, 9]] * B
    assert parent.cpu().tolist() == [0, 0, 0, 3, 3, 3]
    assert nx[0].cpu().tolist() == [-1.0, -2.0, -3.0] * B
    # step does not fit the token buffer -> rejected, nothing launched
    assert L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tv.data_ptr(), ti.data_ptr(), parent.data_ptr(), None, sqrt_tab.data_ptr(),
                                 B, K, K, t_cap - 1, t_cap, 2, 0, L.stream_ptr()) == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,H", [(5, 64, 64), (768, 512, 512), (33, 1024, 512), (100, 256, 256), (16, 128, 1024), (48, 256, 128)])
def test_linear_res_ln_fused(L, W, dtype, M, K, H):
    """y = LN(x W^T + b + res): the fused skinny kernel against torch, residual aliased with the output."""
    g = torch.Generator().manual_seed(M + K + H)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(H, K, generator=g) / math.sqrt(K)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    b, res = torch.randn(H, generator=g), torch.randn(M, H, generator=g)
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    assert L.lib.mic_linear_res_ln_supported(H) == 1 and L.lib.mic_linear_res_ln_supported(192) == 0
    xd, bd, gd, bed = x.to(DEV, dtype), b.to(DEV), gam.to(DEV), bet.to(DEV)
    wrow = w.to(DEV, dtype).contiguous()
    wd = torch.empty_like(wrow)  # fragment order, written by the library's own packer
    L.check(L.lib.mic_pack_weight_frag(wrow.data_ptr(), wd.data_ptr(), L.dtype_code(dtype), H, K, L.stream_ptr()))
    assert torch.equal(wd.view(-1), W.frag_matrix(w, dtype, DEV).view(-1))  # same layout as the host-side packer
    y = res.to(DEV).clone()  # in place: res == y
    yt = torch.empty(M, H, dtype=dtype, device=DEV)
    L.check(L.lib.mic_linear_res_ln(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), H, gd.data_ptr(), bed.data_ptr(),
                                    y.data_ptr(), yt.data_ptr(), H, L.dtype_code(dtype), M, K, H, L.stream_ptr()))
    ref = F.layer_norm(x @ w.t() + b + res, (H,), gam, bet, 1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), **_tol(dtype))
    np.testing.assert_allclose(yt.float().cpu().numpy(), ref.numpy(), **(_tol(dtype) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)))
    assert L.lib.mic_linear_res_ln(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), H, gd.data_ptr(), bed.data_ptr(),
                                   y.data_ptr(), yt.data_ptr(), H, L.dtype_code(dtype), M, K, 192, L.stream_ptr()) == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,relu,out_bf16", [(768, 1536, 512, 0, 0), (768, 3500, 512, 0, 0), (5, 200, 64, 0, 0), (100, 1024, 512, 1, 1),
                                                 (33, 130, 128, 1, 0), (2048, 512, 256, 0, 0)])
def test_linear_skinny(L, W, dtype, M, N, K, relu, out_bf16):
    g = torch.Generator().manual_seed(M * 3 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    b = torch.randn(N, generator=g)
    wf = W.frag_matrix(W._pad_rows(w), dtype, DEV)
    xd, bd = x.to(DEV, dtype), b.to(DEV)
    odt = torch.bfloat16 if out_bf16 else torch.float32
    ldc = (N + 7) // 8 * 8 + 8  # a row stride larger than N
    out = torch.full((M, ldc), 7.0, dtype=odt, device=DEV)
    L.check(L.lib.mic_linear_skinny(xd.data_ptr(), K, wf.data_ptr(), bd.data_ptr(), out.data_ptr(), L.dtype_code(odt), ldc,
                                    L.dtype_code(dtype), M, N, K, relu, L.stream_ptr()))
    ref = x @ w.t() + b
    if relu:
        ref = F.relu(ref)
    tol = _tol(dtype) if not out_bf16 else dict(rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(out[:, :N].float().cpu().numpy(), ref.numpy(), **tol)
    assert (out[:, N:].float() == 7.0).all()  # nothing written past column N


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(768, 512), (37, 64), (100, 320), (16, 1024)])
def test_skinny_layernorm_chain(L, W, dtype, M, H):
    """Two chained launches reproduce GEMM -> +residual -> LayerNorm -> GEMM:
       y, stats = x0 W1^T + b1 + r   (statistics emitted per 128-column part)
       z        = LN(y) W2^T + b2 + LN(y)   (input and residual both LayerNorm views of y, in place)."""
    g = torch.Generator().manual_seed(M + H)

    def rnd(*s, scale=1.0):
        t = torch.randn(*s, generator=g) * scale
        return t.to(dtype).float() if dtype == torch.bfloat16 else t

    x0, w1, w2 = rnd(M, H), rnd(H, H, scale=H ** -0.5), rnd(H, H, scale=H ** -0.5)
    b1, b2, r = torch.randn(H, generator=g), torch.randn(H, generator=g), torch.randn(M, H, generator=g) * 2 + 0.5
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    parts = (H + 127) // 128
    code = L.dtype_code(dtype)
    keep = dict(x0=x0.to(DEV, dtype), w1=W.frag_matrix(W._pad_rows(w1), dtype, DEV), w2=W.frag_matrix(W._pad_rows(w2), dtype, DEV),
                b1=b1.to(DEV), b2=b2.to(DEV), r=r.to(DEV), g=gam.to(DEV), be=bet.to(DEV),
                y=torch.empty(M, H, device=DEV), st=torch.zeros(2, M, parts, 2, device=DEV))
    d = L.SkinnyDesc()
    d.x, d.ld_x, d.wf, d.bias, d.res, d.ld_res = keep["x0"].data_ptr(), H, keep["w1"].data_ptr(), keep["b1"].data_ptr(), keep["r"].data_ptr(), H
    d.out, d.stats_out, d.dtype, d.out_dtype, d.ldc, d.M, d.N, d.K = keep["y"].data_ptr(), keep["st"][0].data_ptr(), code, L.F32, H, M, H, H
    L.check(L.lib.mic_linear_skinny_ex(C.byref(d), L.stream_ptr()))
    y_ref = x0 @ w1.t() + b1 + r
    np.testing.assert_allclose(keep["y"].cpu().numpy(), y_ref.numpy(), **_tol(dtype))
    st = keep["st"][0].cpu()
    for p in range(parts):
        sl = y_ref[:, 128 * p: 128 * (p + 1)]
        np.testing.assert_allclose(st[:, p, 0].numpy(), sl.mean(1).numpy(), rtol=1e-3, atol=2e-3)
        np.testing.assert_allclose(st[:, p, 1].numpy(), ((sl - sl.mean(1, keepdim=True)) ** 2).sum(1).numpy(), rtol=2e-3, atol=2e-3)
    # second launch: input and residual are LayerNorm views of y; output written over y
    view = L.LnRef(keep["y"].data_ptr(), keep["st"][0].data_ptr(), keep["g"].data_ptr(), keep["be"].data_ptr(), H, parts, H)
    d2 = L.SkinnyDesc()
    d2.x_ln, d2.res_ln, d2.wf, d2.bias = view, view, keep["w2"].data_ptr(), keep["b2"].data_ptr()
    d2.out, d2.stats_out, d2.dtype, d2.out_dtype, d2.ldc, d2.M, d2.N, d2.K = keep["y"].data_ptr(), keep["st"][1].data_ptr(), code, L.F32, H, M, H, H
    L.check(L.lib.mic_linear_skinny_ex(C.byref(d2), L.stream_ptr()))
    ln = F.layer_norm(y_ref, (H,), gam, bet, 1e-5)
    ln_in = ln.to(dtype).float() if dtype == torch.bfloat16 else ln
    z_ref = ln_in @ w2.t() + b2 + ln
    np.testing.assert_allclose(keep["y"].cpu().numpy(), z_ref.numpy(), **(_tol(dtype) if dtype == torch.float32 else dict(rtol=3e-2, atol=5e-2)))
    # rejected: view width that does not match K
    bad = L.LnRef(keep["y"].data_ptr(), keep["st"][0].data_ptr(), keep["g"].data_ptr(), keep["be"].data_ptr(), H, parts, H + 4)
    d2.x_ln = bad
    assert L.lib.mic_linear_skinny_ex(C.byref(d2), L.stream_ptr()) == -1
