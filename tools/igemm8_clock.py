#!/usr/bin/env python3
"""In-kernel clock of the eight-phase convolution variant's main loop (measurement build MIC_IGEMM8_DBG=3): cycle counter
over the 100 MHz counter, per workgroup, after a second of back-to-back launches on random data."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from igemm8_check import DEV, L  # noqa: E402
from igemm8_ab import SHAPES  # noqa: E402

os.environ["MIC_IGEMM8_DBG"] = "3"
for name in sys.argv[1:] or ["s2c2", "g4k"]:
    B, H, Ci, Co, k, s, _ = SHAPES[name]
    p = 1 if k == 3 else 0
    OH = (H + 2 * p - k) // s + 1
    x = (torch.rand(B, H, H, Ci, device=DEV) * 2 - 1).to(torch.bfloat16)
    wm = ((torch.rand(Co, k * k * Ci, device=DEV) * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(Co, device=DEV)
    out = torch.empty(B, OH, OH, Co, dtype=torch.bfloat16, device=DEV)
    M, K = B * OH * OH, k * k * Ci
    grid = (M + 255) // 256 * (Co // 256)
    trace = torch.zeros(grid * 8, dtype=torch.int64, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out, d.ln_out = x.data_ptr(), wm.data_ptr(), bias.data_ptr(), out.data_ptr(), trace.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K = M, Co, K
    d.OH, d.OW, d.IH, d.IW = OH, OH, H, H
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = H * H * Ci, H * Ci, Ci
    d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = k, Ci, s, p, p
    d.ldc, d.ld_res, d.relu, d.w_rows = Co, Co, 0, wm.shape[0]
    t0 = time.time()
    n = 0
    while time.time() - t0 < 1.5:
        for _ in range(50):
            L.check(L.lib.mic_conv2d_variant(C.byref(d), 18, L.stream_ptr()), "conv")
        torch.cuda.synchronize()
        n += 50
    t = trace.cpu().numpy().reshape(grid, 8)
    cyc = (t[:, 1] - t[:, 0]).astype(np.float64)
    rt = (t[:, 3] - t[:, 2]).astype(np.float64)
    ghz = cyc / rt * 0.1
    nk = K // 64
    print(f"{name}: {grid} workgroups, {nk} K-tiles; loop {np.median(rt) / 100:.1f} us, {np.median(cyc) / nk:.0f} cycles per K-tile "
          f"(2048 = MFMA-bound), clock median {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f}); "
          f"entry to loop {np.median(t[:, 2] - t[:, 4]) / 100:.1f} us, loop end to last store {np.median(t[:, 5] - t[:, 3]) / 100:.1f} us, "
          f"first entry to last exit {(t[:, 5].max() - t[:, 4].min()) / 100:.1f} us, entry spread {(t[:, 4].max() - t[:, 4].min()) / 100:.1f} us", flush=True)
