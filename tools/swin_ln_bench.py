#!/usr/bin/env python3
"""Times the Swin-V2 projection / MLP-2 GEMMs with LayerNorm in the epilogue against GEMM + stand-alone LayerNorm (development aid)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_captioning_amd as amd

L, W = amd._lib, amd.weights
DEV = "cuda:0"
B = int(os.environ.get("B", 512))


def desc(x, wm, bias, out, N):
    M, K = x.shape
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out = x.data_ptr(), wm.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.dtype, d.out_dtype, d.res_dtype = L.dtype_code(x.dtype), L.F32, L.F32
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, N, K, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride = K, 4, 1, K, 1
    d.ldc, d.ld_res, d.w_rows = N, N, wm.shape[0]
    return d


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for toks, Cc in ((3136, 96), (784, 192), (196, 384)):
    for K in (Cc, 4 * Cc):
        M = B * toks
        x = torch.randn(M, K, device=DEV).bfloat16()
        wm = W.conv_matrix((torch.randn(Cc, K) / K ** 0.5).view(Cc, K, 1, 1), torch.bfloat16, DEV)
        bias, ga, be = (torch.randn(Cc, device=DEV) for _ in range(3))
        y, u, t = torch.randn(M, Cc, device=DEV), torch.empty(M, Cc, device=DEV), torch.empty(M, Cc, device=DEV, dtype=torch.bfloat16)
        d2 = desc(x, wm, bias, u, Cc)
        d1 = desc(x, wm, bias, y, Cc)
        d1.res = y.data_ptr()
        d1.ln_gamma, d1.ln_beta, d1.ln_out, d1.ld_ln_out = ga.data_ptr(), be.data_ptr(), t.data_ptr(), Cc
        s = L.stream_ptr()

        def two():
            L.check(L.lib.mic_conv2d(C.byref(d2), s))
            L.check(L.lib.mic_layernorm_add(u.data_ptr(), ga.data_ptr(), be.data_ptr(), y.data_ptr(), y.data_ptr(), t.data_ptr(), Cc, L.BF16, M, Cc, s))

        def gemm_only():
            L.check(L.lib.mic_conv2d(C.byref(d2), s))

        def one():
            L.check(L.lib.mic_conv2d(C.byref(d1), s))

        t2, tg, t1 = timeit(two), timeit(gemm_only), timeit(one)
        fl = 2.0 * M * Cc * K
        print(f"M={M:8d} N={Cc:4d} K={K:5d}  gemm {tg:7.1f} us ({fl / tg / 1e6:6.1f} TF)  gemm+ln {t2:7.1f} us  fused {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF)", flush=True)
