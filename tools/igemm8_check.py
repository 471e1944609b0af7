#!/usr/bin/env python3
"""The 256x256 eight-phase convolution variant (18) against the other tile variants: results on small shapes with row,
tap and K-tile-count edge cases (against torch's fp32 convolution and bit for bit against variant 0), then HIP-event
timings on the ResNet-50 layer shapes it is meant for (batch 256), variant 18 beside the variant the heuristic picks."""
import ctypes as C
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L, W = amd._lib, amd.weights
DEV = "cuda:0"


_SCRATCH = {}


def scratch():
    """One zeroed split-K scratch per process (the launches leave its tickets zero)."""
    if "t" not in _SCRATCH:
        _SCRATCH["t"] = torch.zeros(int(L.lib.mic_conv2d_scratch_bytes()), dtype=torch.uint8, device=DEV)
    return _SCRATCH["t"]


def conv(variant, x, wm, bias, res, B, Ci, H, Wd, Co, k, stride, pad, relu=1, split=True):
    OH, OW = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    out = torch.empty(B, OH, OW, Co, dtype=torch.bfloat16, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.res, d.out = x.data_ptr(), wm.data_ptr(), bias.data_ptr(), L.ptr(res), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K = B * OH * OW, Co, k * k * Ci
    d.OH, d.OW, d.IH, d.IW = OH, OW, H, Wd
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = H * Wd * Ci, Wd * Ci, Ci
    d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = k, Ci, stride, pad, pad
    d.ldc, d.ld_res, d.relu, d.w_rows = Co, Co, relu, wm.shape[0]
    if split:
        d.scratch, d.scratch_bytes = scratch().data_ptr(), scratch().numel()
    if variant < 0:
        L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()), "mic_conv2d")
    else:
        L.check(L.lib.mic_conv2d_variant(C.byref(d), variant, L.stream_ptr()), f"mic_conv2d_variant({variant})")
    return out, d


CASES = [  # B, Ci, H, W, Co, k, stride, pad, residual
    (3, 64, 13, 11, 256, 1, 1, 0, True),     # one K-tile, M = 429: a partial second row tile
    (2, 64, 14, 14, 512, 3, 1, 1, False),    # 9 K-tiles (odd), taps outside the image
    (2, 128, 15, 15, 256, 3, 2, 1, False),   # stride 2, 18 K-tiles
    (1, 192, 9, 9, 256, 1, 1, 0, True),      # 3 K-tiles, M = 81
    (5, 256, 14, 14, 768, 3, 1, 1, True),    # several tiles in both directions, 36 K-tiles (with a scratch: one slice; four of 9 under MIC_IGEMM8_SPLITK=4)
    (3, 512, 7, 7, 256, 3, 1, 1, False),     # one tile, 72 K-tiles: three slices of 24
    (2, 320, 9, 9, 512, 1, 1, 0, True),      # 5 K-tiles: too short to split
]


def check():
    ok = True
    for case in CASES:
        B, Ci, H, Wd, Co, k, s, p, use_res = case
        g = torch.Generator().manual_seed(sum(case))
        x = torch.randn(B, Ci, H, Wd, generator=g).bfloat16().float()
        w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).bfloat16().float()
        b = torch.randn(Co, generator=g)
        ref = F.conv2d(x, w, b, s, p)
        res = None
        if use_res:
            res = torch.randn(ref.shape, generator=g).bfloat16().float()
            ref = ref + res
        ref = F.relu(ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
        wm = W.conv_matrix(w, torch.bfloat16, DEV)
        rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
        bd = b.to(DEV)
        o18, _ = conv(18, xd, wm, bd, rd, B, Ci, H, Wd, Co, k, s, p, split=False)
        o18s, _ = conv(18, xd, wm, bd, rd, B, Ci, H, Wd, Co, k, s, p, split=True)
        o18s2, _ = conv(18, xd, wm, bd, rd, B, Ci, H, Wd, Co, k, s, p, split=True)
        o0, _ = conv(0, xd, wm, bd, rd, B, Ci, H, Wd, Co, k, s, p)
        torch.cuda.synchronize()
        got = o18.float().cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs().max().item()
        same = torch.equal(o18, o0)
        close = torch.allclose(got, ref, rtol=2e-2, atol=2e-2)
        # with a scratch the layer may split K: regrouped fp32 sum, the bf16 result within one rounding step of the unsplit one
        d_split = (o18s.float() - o18.float()).abs().max().item()
        split_ok = torch.allclose(o18s.float(), o18.float(), rtol=8e-3, atol=1e-3) and torch.equal(o18s, o18s2)
        tickets_zero = int(scratch()[:4096].to(torch.int32).sum().item()) == 0
        print(f"case {case}: max |err| vs fp32 {err:.4f}  close={close}  bit-equal to variant 0: {same}  "
              f"with scratch: max |diff| {d_split:.4f} ok={split_ok} tickets zero={tickets_zero}", flush=True)
        ok = ok and close and same and split_ok and tickets_zero
    return ok


SHAPES = [  # name, B, H, Cin, Cout, k, stride, residual
    ("s2b0c2", 256, 28, 256, 256, 3, 2, False),
    ("s2b1c1", 256, 14, 1024, 256, 1, 1, False),
    ("s2b1c2", 256, 14, 256, 256, 3, 1, False),
    ("s2b0ds", 256, 28, 512, 1024, 1, 2, False),
    ("s2b1c3+res", 256, 14, 256, 1024, 1, 1, True),
    ("s3b0c1", 256, 14, 1024, 512, 1, 1, False),
    ("s3b0c2", 256, 14, 512, 512, 3, 2, False),
    ("s3b0ds", 256, 14, 1024, 2048, 1, 2, False),
    ("s3b1c1", 256, 7, 2048, 512, 1, 1, False),
    ("s3b1c2", 256, 7, 512, 512, 3, 1, False),
    ("s3b1c3+res", 256, 7, 512, 2048, 1, 1, True),
    ("s1b0ds", 256, 56, 256, 512, 1, 2, False),
    ("s1b1c3+res", 256, 28, 128, 512, 1, 1, True),
    ("s2b0c1", 256, 28, 512, 256, 1, 1, False),
    ("s0b0ds", 256, 56, 64, 256, 1, 1, False),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench(variants):
    for name, B, H, Ci, Co, k, s, use_res in SHAPES:
        p = 1 if k == 3 else 0
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(B, H, H, Ci, device=DEV).to(torch.bfloat16)
        wm = (torch.randn((Co + 127) // 128 * 128, k * k * Ci, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(Co, device=DEV)
        res = torch.randn(B, OH, OH, Co, device=DEV).to(torch.bfloat16) if use_res else None
        M, K = B * OH * OH, k * k * Ci
        line = f"{name:11s} M={M:7d} N={Co:5d} K={K:5d}"
        outs = {}
        for v in variants:
            try:
                us = timeit(lambda: conv(v, x, wm, bias, res, B, Ci, H, H, Co, k, s, p))
                outs[v] = conv(v, x, wm, bias, res, B, Ci, H, H, Co, k, s, p)[0]
                line += f"   v{v}: {us:7.1f} us {2.0 * M * Co * K / us / 1e6:7.1f} TF"
            except RuntimeError as e:
                line += f"   v{v}: refused"
        torch.cuda.synchronize()
        vs = [v for v in variants if v in outs]
        same = all(torch.equal(outs[vs[0]], outs[v]) for v in vs[1:])
        print(line + f"   identical: {same}", flush=True)


if __name__ == "__main__":
    if not check():
        sys.exit("igemm8_check: results differ")
    bench([int(v) for v in sys.argv[1:]] or [-1, 18])
