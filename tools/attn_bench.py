#!/usr/bin/env python3
"""Times the single-query attention kernels at the bench shape (N=768 rows, H=512, 2 heads, bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L = amd._lib
DEV = "cuda:0"
N, H, heads, Tc, S, K = 768, 512, 2, 31, 49, 3
dt = torch.bfloat16


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


qkv = torch.randn(N, 3 * H, device=DEV)
kc = torch.randn(N, Tc, H, device=DEV).to(dt)
vc = torch.randn(N, Tc, H, device=DEV).to(dt)
anc = torch.randint(0, N, (N, Tc), device=DEV, dtype=torch.int32)
o = torch.empty(N, H, device=DEV, dtype=dt)
for t in (0, 1, 7, 15, 29):
    us = timeit(lambda: L.check(L.lib.mic_self_attn_step(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), anc.data_ptr(), t, Tc, o.data_ptr(), H,
                                                         L.BF16, N, H, heads, L.stream_ptr())))
    print(f"self  t={t:2d}: {us:6.1f} us")
kv = torch.randn(N // K, S, 2 * H, device=DEV).to(dt)
q = torch.randn(N, H, device=DEV)
us = timeit(lambda: L.check(L.lib.mic_cross_attn_step(q.data_ptr(), H, kv.data_ptr(), S, K, o.data_ptr(), H, None, 0, 1.0, 0, L.BF16, N, H, heads,
                                                      L.stream_ptr())))
print(f"cross S={S}: {us:6.1f} us")
logits = torch.randn(N, 3500, device=DEV)
tv = torch.empty(N, 3, device=DEV)
ti = torch.empty(N, 3, device=DEV, dtype=torch.int32)
us = timeit(lambda: L.check(L.lib.mic_logsoftmax_topk(logits.data_ptr(), 3500, N, 3500, 3, 1.0, 0, tv.data_ptr(), ti.data_ptr(), None, L.stream_ptr())))
print(f"topk: {us:6.1f} us")
