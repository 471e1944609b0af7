#!/usr/bin/env python3
"""Times mic_conv2d_variant on the GEMM shapes of the Swin-V2-S encoder at batch 512, 224x224 (tokens x channels per
stage: qkv, proj, fc1, fc2, patch merging), every tile variant, HIP events around 10 launches.  Development aid for the
tile heuristic."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L, DEV, dt = amd._lib, "cuda:0", torch.bfloat16
B = int(os.environ.get("B", 512))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,3,7,8,9,10,11,12,13,14,15,16,17").split(",")]


def shapes():
    out = []
    for si, (tok, c) in enumerate([(3136, 96), (784, 192), (196, 384), (49, 768)]):
        M = B * tok
        out += [(f"s{si} qkv", M, 3 * c, c, 0), (f"s{si} proj", M, c, c, 0), (f"s{si} fc1", M, 4 * c, c, 2), (f"s{si} fc2", M, c, 4 * c, 0)]
        if si < 3:
            out.append((f"s{si} merge", M // 4, 2 * c, 4 * c, 0))
    return out


def run(M, N, K, act, variant):
    x = torch.randn(M, K, device=DEV).to(dt)
    w = (torch.randn((N + 127) // 128 * 128, K, device=DEV) * 0.05).to(dt)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=dt)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, N, K, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc, d.relu = K, 4, 1, K, 1, N, act
    d.w_rows = w.shape[0]
    call = (lambda: L.lib.mic_conv2d(C.byref(d), L.stream_ptr())) if variant < 0 else (lambda: L.lib.mic_conv2d_variant(C.byref(d), variant, L.stream_ptr()))
    for _ in range(2):
        if call() != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


for name, M, N, K, act in shapes():
    res = {v: run(M, N, K, act, v) for v in [-1] + VARIANTS}
    best = min((t, v) for v, t in res.items() if t is not None and v >= 0)
    by = 2.0 * (M * K + M * N)
    print(f"{name:9s} M={M:8d} N={N:5d} K={K:5d} auto {res[-1]:7.1f} us  best v{best[1]:<2d} {best[0]:7.1f} us ({2.0 * M * N * K / best[0] / 1e6:6.1f} TF, {by / best[0] / 1e6:4.2f} TB/s)  "
          + " ".join(f"v{v}:{t:.0f}" for v, t in res.items() if t is not None and v >= 0), flush=True)
