#!/usr/bin/env python3
"""Times mic_conv2d on every convolution shape of the ResNet-50 encoder (batch 256, 224x224) and
on the decode-step GEMM shapes, one by one, with HIP events.  Development aid for tile tuning."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L = amd._lib
DEV = "cuda:0"


def layers(B=256):
    out = [("stem", B, 230, 232, 4, 64, 7, 1, 2, 0, 32, 147)]  # name,B,IH,IW,Cpix,Cout,kh,kw,stride,pad,run,k_valid
    cin, h = 64, 56
    for si, (w, nb, st) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(nb):
            s = st if b == 0 else 1
            out.append((f"s{si}b{b}c1", B, h, h, cin, w, 1, 1, 1, 0, cin, 0))
            out.append((f"s{si}b{b}c2", B, h, h, w, w, 3, 3, s, 1, w, 0))
            if b == 0:
                out.append((f"s{si}b{b}ds", B, h, h, cin, 4 * w, 1, 1, s, 0, cin, 0))
            out.append((f"s{si}b{b}c3+res", B, h // s, h // s, w, 4 * w, 1, 1, 1, 0, w, 0))
            cin, h = 4 * w, h // s
    out.append(("proj", B, 7, 7, 2048, 512, 1, 1, 1, 0, 2048, 0))
    return out


def run(desc, iters=10, variant=-1):
    def call():
        if variant < 0:
            L.check(L.lib.mic_conv2d(C.byref(desc), L.stream_ptr()))
        else:
            L.check(L.lib.mic_conv2d_variant(C.byref(desc), variant, L.stream_ptr()))
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    global SCRATCH
    dt = torch.bfloat16
    SCRATCH = torch.zeros(int(L.lib.mic_conv2d_scratch_bytes()), dtype=torch.uint8, device=DEV)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    tot = 0.0
    seen = {}
    for (name, B, IH, IW, cpix, cout, kh, kw, stride, pad, run_, kv) in layers():
        key = (IH, IW, cpix, cout, kh, stride, "res" in name)
        if only and only not in name:
            continue
        stem = name == "stem"
        OH = (IH - 7) // 2 + 1 if stem else (IH + 2 * pad - kh) // stride + 1
        OW = 112 if stem else (IW + 2 * pad - kw) // stride + 1
        K = kh * kw * run_
        M = B * OH * OW
        if key in seen:
            us = seen[key]
        else:
            x = torch.randn(B, IH, IW, cpix, device=DEV).to(dt)
            w = (torch.randn((cout + 127) // 128 * 128, K, device=DEV) * 0.05).to(dt)
            bias = torch.randn(cout, device=DEV)
            res = torch.randn(M, cout, device=DEV).to(dt) if "res" in name else None
            out = torch.empty(M, cout, device=DEV, dtype=dt)
            d = L.ConvDesc()
            d.in_, d.w, d.bias, d.res, d.out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), L.ptr(res), out.data_ptr()
            d.dtype = d.out_dtype = d.res_dtype = L.BF16
            d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, cout, K, OH, OW, IH, IW
            d.in_img_stride, d.in_row_stride, d.in_pix_stride = IH * IW * cpix, IW * cpix, cpix
            d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = kw, run_, stride, pad, pad
            d.ldc, d.ld_res, d.relu = cout, cout, 1
            d.scratch, d.scratch_bytes = SCRATCH.data_ptr(), SCRATCH.numel()  # as in the trunk: late layers may split K
            us = run(d)
            seen[key] = us
            del x, w, res, out
        fl = 2.0 * M * cout * (kv or K)
        by = 2.0 * (M * (kv or K) / (kh * kw if not stem else 1) * (1 if stride == 1 else 1) + M * cout * (2 if "res" in name else 1))
        tot += us
        print(f"{name:12s} M={M:8d} N={cout:5d} K={K:5d} {us:8.1f} us {fl / us / 1e6:7.1f} TF  ~{by / us / 1e6:5.2f} TB/s")
    print(f"encoder GEMM total {tot / 1e3:.2f} ms")
    if only:
        return
    for (M, N, K, nm) in [(768, 1536, 512, "qkv"), (768, 512, 512, "proj"), (768, 1024, 512, "ff1"), (768, 512, 1024, "ff2"),
                          (768, 3500, 512, "vocab"), (12544, 1024, 512, "cross_kv")]:
        x = torch.randn(M, K, device=DEV).to(dt)
        w = (torch.randn((N + 127) // 128 * 128, K, device=DEV) * 0.05).to(dt)
        out = torch.empty(M, N, device=DEV)
        d = L.ConvDesc()
        d.in_, d.w, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
        d.dtype, d.out_dtype = L.BF16, L.F32
        d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, N, K, 1, 1, 1, 1
        d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc = K, 4, 1, K, 1, N
        us = run(d, 50)
        print(f"{nm:12s} M={M:8d} N={N:5d} K={K:5d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF")


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("ln", "skinny", "tail")):
    main()


def fused_ln():
    dt = torch.bfloat16
    for (M, K, H) in [(768, 512, 512), (768, 1024, 512)]:
        x = torch.randn(M, K, device=DEV).to(dt)
        w = amd.weights.frag_matrix(torch.randn(H, K) * 0.05, dt, DEV)
        b, g, be = torch.randn(H, device=DEV), torch.randn(H, device=DEV), torch.randn(H, device=DEV)
        y = torch.randn(M, H, device=DEV)
        yt = torch.empty(M, H, device=DEV, dtype=dt)

        def call():
            L.check(L.lib.mic_linear_res_ln(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), H, g.data_ptr(), be.data_ptr(),
                                            y.data_ptr(), yt.data_ptr(), H, L.BF16, M, K, H, L.stream_ptr()))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"gemm_res_ln  M={M} K={K} H={H}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ln":
    fused_ln()


def skinny():
    dt = torch.bfloat16
    for (M, N, K, nm) in [(768, 512, 512, "tail"), (768, 1536, 512, "qkv"), (768, 1024, 512, "ff1"), (768, 512, 1024, "ff2-tail"), (768, 3500, 512, "vocab")]:
        x = torch.randn(M, K, device=DEV).to(dt)
        w = amd.weights.frag_matrix(amd.weights._pad_rows(torch.randn(N, K) * 0.05), dt, DEV)
        b = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        st = torch.zeros(M, (N + 127) // 128, 2, device=DEV)
        res = torch.randn(M, N, device=DEV)
        d = L.SkinnyDesc()
        d.x, d.ld_x, d.wf, d.bias, d.out = x.data_ptr(), K, w.data_ptr(), b.data_ptr(), out.data_ptr()
        d.dtype, d.out_dtype, d.ldc, d.M, d.N, d.K = L.BF16, L.F32, N, M, N, K
        if "tail" in nm:
            d.res, d.ld_res, d.stats_out = res.data_ptr(), N, st.data_ptr()

        def call():
            L.check(L.lib.mic_linear_skinny_ex(C.byref(d), L.stream_ptr()))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"skinny {nm:9s} M={M} N={N} K={K}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "skinny":
    skinny()


def tails():
    """conv3 + shortcut convolution of each stage's first bottleneck as one two-segment GEMM."""
    dt, B = torch.bfloat16, 256
    cin, h = 64, 56
    for si, (wd, st) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)]):
        oh = h // st
        M, N, K1, K2 = B * oh * oh, 4 * wd, wd, cin
        y2 = torch.randn(M, K1, device=DEV).to(dt)
        x = torch.randn(B * h * h, K2, device=DEV).to(dt)
        w = (torch.randn((N + 127) // 128 * 128, K1 + K2, device=DEV) * 0.05).to(dt)
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV, dtype=dt)
        d = L.ConvDesc()
        d.in_, d.w, d.bias, d.out = y2.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr()
        d.dtype = d.out_dtype = d.res_dtype = L.BF16
        d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = M, N, K1, oh, oh, oh, oh
        d.in_img_stride, d.in_row_stride, d.in_pix_stride = oh * oh * K1, oh * K1, K1
        d.kw_taps, d.run, d.stride = 1, K1, 1
        d.ldc, d.relu = N, 1
        d.in2, d.K2, d.stride2 = x.data_ptr(), K2, st
        d.in2_img_stride, d.in2_row_stride, d.in2_pix_stride = h * h * K2, h * K2, K2
        scratch = torch.zeros(int(L.lib.mic_conv2d_scratch_bytes()), dtype=torch.uint8, device=DEV)
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
        fl = 2.0 * M * N * (K1 + K2)
        by = 2.0 * (M * K1 + M * K2 + M * N)
        line = f"s{si} tail M={M:8d} N={N:5d} K={K1}+{K2}"
        for v in (-1, 0, 11, 16, 18):  # the heuristic's pick, then the variants that take two K segments
            try:
                us = run(d, variant=v)
            except L.MicError:
                continue
            line += f" | {v}: {us:7.1f} us {fl / us / 1e6:6.1f} TF ~{by / us / 1e6:4.2f} TB/s"
        print(line)
        cin, h = 4 * wd, oh


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tail":
    tails()
    sys.exit(0)
