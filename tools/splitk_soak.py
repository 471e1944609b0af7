#!/usr/bin/env python3
"""Soak run of the in-launch split-K reduction of the 256x256 eight-phase convolution variant: the ResNet-50 stage-3 3x3
layer at batch 256 (98 tiles x 2 slices, K = 4608) launched many times on one scratch; every output must equal the first
launch's bit for bit (the last arriver adds the slices in slice order, so the sum does not depend on who arrives last), the
tickets must be zero after every launch, and the first output must match a launch that does not split K within bf16 rounding.
usage: splitk_soak.py [launches]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from igemm8_check import conv, scratch, DEV, L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    B, H, Ci, Co, k = 256, 7, 512, 512, 3
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, H, H, Ci, generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    wm = ((torch.rand(Co, k * k * Ci, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).to(DEV)
    bias = torch.randn(Co, generator=g).to(DEV)
    first, _ = conv(18, x, wm, bias, None, B, Ci, H, H, Co, k, 1, 1)
    plain, _ = conv(18, x, wm, bias, None, B, Ci, H, H, Co, k, 1, 1, split=False)
    torch.cuda.synchronize()
    rel = ((first.float() - plain.float()).abs().max() / plain.float().abs().max()).item()
    bad = dirty = 0
    t0 = time.time()
    for i in range(n):
        out, _ = conv(18, x, wm, bias, None, B, Ci, H, H, Co, k, 1, 1)
        bad += not torch.equal(out, first)
        dirty += int(scratch()[:4096].any().item())
        if (i + 1) % 500 == 0:
            print(f"launch {i + 1}/{n}: {bad} differing outputs, {dirty} launches left a ticket set, {time.time() - t0:.0f} s", flush=True)
    print(f"{n} split-K launches (98 tiles x 2 slices): {bad} outputs differ from the first, {dirty} left a ticket set; "
          f"split against unsplit: max difference {rel:.2e} of the largest output")
    return 1 if bad or dirty or rel > 2e-2 else 0


if __name__ == "__main__":
    sys.exit(main())
