#!/usr/bin/env python3
"""Interleaved timing rounds of convolution tile variants / measurement builds in ONE process on one device.
usage: igemm8_ab.py [--rounds N] [--shapes a,b] CONFIG...   with CONFIG = variant[:ENV=VAL[:ENV=VAL...]], e.g.
  16  18  18:MIC_IGEMM8_LINES=1  18:MIC_IGEMM8_DBG=2"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ctypes as C  # noqa: E402

from igemm8_check import conv, DEV, L  # noqa: E402

SHAPES = {  # name: B, H, Cin, Cout, k, stride, residual
    "s2c2": (256, 14, 256, 256, 3, 1, False), "s2c1": (256, 14, 1024, 256, 1, 1, False), "s2ds": (256, 28, 512, 1024, 1, 2, False),
    "s2c3": (256, 14, 256, 1024, 1, 1, True), "s3c1a": (256, 14, 1024, 512, 1, 1, False), "s3c2": (256, 7, 512, 512, 3, 1, False),
    "s3ds": (256, 14, 1024, 2048, 1, 2, False), "s3c1": (256, 7, 2048, 512, 1, 1, False), "s3c3": (256, 7, 512, 2048, 1, 1, True),
    "s2c1a": (256, 28, 512, 256, 1, 1, False), "s1c2": (256, 28, 128, 128, 3, 1, False), "g4k": (1, 64, 4096, 4096, 1, 1, False),
    "s1c1": (256, 28, 512, 128, 1, 1, False), "s0c1": (256, 56, 256, 64, 1, 1, False), "s1ds": (256, 56, 256, 512, 1, 2, False),
    "s1c3": (256, 28, 128, 512, 1, 1, True), "s0c2": (256, 56, 64, 64, 3, 1, False), "s0c3": (256, 56, 64, 256, 1, 1, True),
    "s1b0c2": (256, 56, 128, 128, 3, 2, False), "s2b0c2": (256, 28, 256, 256, 3, 2, False), "s3b0c2": (256, 14, 512, 512, 3, 2, False),
    "ckv": (256, 7, 512, 1024, 1, 1, False), "ckv6": (256, 7, 512, 6144, 1, 1, False),
    # Swin-V2 GEMMs without the LayerNorm epilogue at batch 512: QKV and MLP-1 of the four stages
    "w0qkv": (512, 56, 96, 288, 1, 1, False), "w0fc1": (512, 56, 96, 384, 1, 1, False),
    "w1qkv": (512, 28, 192, 576, 1, 1, False), "w1fc1": (512, 28, 192, 768, 1, 1, False),
    "w2qkv": (512, 14, 384, 1152, 1, 1, False), "w2fc1": (512, 14, 384, 1536, 1, 1, False),
    "w3qkv": (512, 7, 768, 2304, 1, 1, False), "w3fc1": (512, 7, 768, 3072, 1, 1, False),
    "tiny": (1, 16, 64, 256, 1, 1, False), "tiny196": (196, 16, 64, 256, 1, 1, False),
    "g8k": (4, 64 * 0 + 45, 8192, 8192, 1, 1, False), "gsk": (16, 64, 2304, 256, 1, 1, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--shapes", default="s2c2,s2c1,s3c2,s3c1a,g4k")
    ap.add_argument("--batch", type=int, default=0, help="images instead of each shape's own batch (256 for the ResNet layers)")
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    cfgs = []
    for c in args.configs:
        parts = c.split(":")
        cfgs.append((c, int(parts[0]), dict(kv.split("=") for kv in parts[1:])))
    envs = sorted({k for _, _, e in cfgs for k in e})
    for name in args.shapes.split(","):
        B, H, Ci, Co, k, s, use_res = SHAPES[name]
        B = args.batch or B
        p = 1 if k == 3 else 0
        OH = (H + 2 * p - k) // s + 1
        x = (torch.rand(B, H, H, Ci, device=DEV) * 2 - 1).to(torch.bfloat16)
        wm = ((torch.rand((Co + 127) // 128 * 128, k * k * Ci, device=DEV) * 2 - 1) * 0.05).to(torch.bfloat16)
        bias = torch.randn(Co, device=DEV)
        res = torch.randn(B, OH, OH, Co, device=DEV).to(torch.bfloat16) if use_res else None
        M, K = B * OH * OH, k * k * Ci
        times = {c: [] for c, _, _ in cfgs}
        for r in range(args.rounds + 1):
            for label, v, env in cfgs:
                for e in envs:
                    os.environ.pop(e, None)
                os.environ.update(env)
                if label not in times:
                    continue
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                try:
                    out, d = conv(v, x, wm, bias, res, B, Ci, H, H, Co, k, s, p)
                except L.MicError:  # a tile variant that does not take this shape
                    del times[label]
                    continue
                ref, st, fn = C.byref(d), L.stream_ptr(), (L.lib.mic_conv2d if v < 0 else L.lib.mic_conv2d_variant)
                ev[0].record()
                for _ in range(args.iters):  # the descriptor is built once: the host must not be what is timed
                    rc = fn(ref, st) if v < 0 else fn(ref, v, st)
                ev[1].record()
                L.check(rc, "mic_conv2d")
                torch.cuda.synchronize()
                if r:  # round 0 warms up
                    times[label].append(ev[0].elapsed_time(ev[1]) / args.iters * 1e3)
        line = f"{name:6s} M={M:7d} N={Co:5d} K={K:5d}"
        for label, _, _ in cfgs:
            if label not in times:
                continue
            ts = sorted(times[label])
            med = ts[len(ts) // 2]
            line += f" | {label}: {med:7.1f} us (min {ts[0]:.1f}) {2.0 * M * Co * K / med / 1e6:6.0f} TF"
        print(line, flush=True)
        for e in envs:
            os.environ.pop(e, None)


if __name__ == "__main__":
    main()
