#!/usr/bin/env python3
"""Times the one-launch beam tail for several vocabulary sizes (development aid: how much of it is the pass over the logits)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L = amd._lib
DEV = "cuda:0"
B, K, t_cap, H, eos, t = 256, int(os.environ.get("K", 3)), 31, 512, 2, 7
N = B * K
g = torch.Generator().manual_seed(0)
sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(DEV)
for V in (3500, 1024, 256, 64):
    logits = (torch.randn(N, V, generator=g) * 3).to(DEV)
    tok_emb, pos_emb = torch.randn(V, H, generator=g).to(DEV), torch.randn(t_cap, H, generator=g).to(DEV)
    bufs = []
    for _ in range(2):
        bufs.append((torch.zeros(N, dtype=torch.float64, device=DEV), torch.full((N,), t + 1, dtype=torch.int32, device=DEV),
                     torch.randint(4, 50, (N, t_cap), generator=g).int().to(DEV), torch.zeros(N, t_cap, dtype=torch.int32, device=DEV)))
    cur, nxt = L.BeamState(), L.BeamState()
    for st, b in zip((cur, nxt), bufs):
        st.score, st.len, st.seq, st.anc = (v.data_ptr() for v in b)
    parent = torch.zeros(N, dtype=torch.int32, device=DEV)
    live = torch.zeros(1, dtype=torch.int32, device=DEV)
    x, xt = torch.zeros(N, H, device=DEV), torch.zeros(N, H, device=DEV, dtype=torch.bfloat16)

    def run():
        L.check(L.lib.mic_beam_tail(logits.data_ptr(), V, V, C.byref(cur), C.byref(nxt), parent.data_ptr(), live.data_ptr(),
                                    sqrt_tab.data_ptr(), B, K, K, t, t_cap, eos, tok_emb.data_ptr(), pos_emb.data_ptr(), x.data_ptr(),
                                    xt.data_ptr(), L.BF16, H, V, L.stream_ptr()))

    def run_noemb():
        L.check(L.lib.mic_beam_tail(logits.data_ptr(), V, V, C.byref(cur), C.byref(nxt), parent.data_ptr(), live.data_ptr(),
                                    sqrt_tab.data_ptr(), B, K, K, t, t_cap, eos, None, None, None, None, L.BF16, H, V, L.stream_ptr()))

    def run_parts():
        tv, ti = run_parts.tv, run_parts.ti
        L.check(L.lib.mic_logsoftmax_topk(logits.data_ptr(), V, N, V, K, 1.0, 0, tv.data_ptr(), ti.data_ptr(), None, L.stream_ptr()))
        L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tv.data_ptr(), ti.data_ptr(), parent.data_ptr(), live.data_ptr(),
                                      sqrt_tab.data_ptr(), B, K, K, t, t_cap, eos, 0, L.stream_ptr()))

    run_parts.tv, run_parts.ti = torch.empty(N, K, device=DEV), torch.empty(N, K, dtype=torch.int32, device=DEV)
    for fn, nm in ((run_noemb, "no embedding"), (run_parts, "top-k + select as two launches")):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"V={V:5d}  {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us  {nm}")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"V={V:5d}  {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per launch (back to back)")
