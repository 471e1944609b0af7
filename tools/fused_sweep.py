"""Times mic_bottleneck_fused at the ResNet-50 shapes of batch 256 (HIP events around 10 launches).
MIC_FUSED_DEBUG=<bits> skips phases of the kernel for timing experiments (results are then wrong)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L = amd._lib
DEV = "cuda:0"
bf = torch.bfloat16


def run(Cm, C2, B, H):
    M = B * H * H
    t = dict(t1=torch.randn(M, Cm, device=DEV).to(bf), w2=torch.randn(128, 9 * Cm, device=DEV).to(bf) * 0.04,
             b2=torch.zeros(Cm, device=DEV), w3=torch.randn(4 * Cm, Cm, device=DEV).to(bf) * 0.1, b3=torch.zeros(4 * Cm, device=DEV),
             res=torch.randn(M, 4 * Cm, device=DEV).to(bf), xout=torch.empty(M, 4 * Cm, device=DEV, dtype=bf),
             w1n=torch.randn(128, 4 * Cm, device=DEV).to(bf) * 0.05, b1n=torch.zeros(128, device=DEV),
             t1n=torch.empty(M, max(C2, 8), device=DEV, dtype=bf))
    d = L.FusedBlockDesc()
    for k, v in t.items():
        if C2 or k not in ("w1n", "b1n", "t1n"):
            setattr(d, k, v.data_ptr())
    d.dtype, d.B, d.H, d.W, d.C, d.C2 = L.BF16, B, H, H, Cm, C2
    d.w2_rows, d.w3_rows, d.w1n_rows = 128, 4 * Cm, 128
    s = L.stream_ptr()
    for _ in range(3):
        L.check(L.lib.mic_bottleneck_fused(C.byref(d), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(L.lib.mic_bottleneck_fused(C.byref(d), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * M * (9 * Cm * Cm + 4 * Cm * Cm + 4 * Cm * C2)
    by = 2.0 * M * (Cm + 8 * Cm + C2)
    print(f"C={Cm:4d} C2={C2:4d} {H}x{H}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF  {by / us / 1e6:5.2f} TB/s  dbg={os.environ.get('MIC_FUSED_DEBUG', '0')}")


if __name__ == "__main__":
    for Cm, C2, H in ((64, 64, 56), (64, 128, 56), (128, 128, 28), (128, 0, 28)):
        run(Cm, C2, 256, H)
