#!/usr/bin/env python3
"""Times mic_attn_fused alone at the headline decode shape (768 rows = 256 images x 3 beams, H=512 bf16, 2 heads,
49 image tokens) for a few cached lengths.  Development aid (R and HEADS in the environment change the shape)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

L, DEV = amd._lib, "cuda:0"
B, R, S, H, heads, t_cap = 256, int(os.environ.get("R", 3)), 49, 512, int(os.environ.get("HEADS", 2)), 31
N = B * R
dt = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
# three layers' worth of buffers, cycled, so that a call does not find its operands in the caches left by the previous one
Ly = 3
qkvf = rn(N, (2 + heads) * H)
kc, vfc = [rn(N, t_cap, H).to(dt) for _ in range(Ly)], [rn(N, t_cap, heads, H).to(dt) for _ in range(Ly)]
anc = (torch.arange(N, device=DEV) // R * R)[:, None].expand(N, t_cap).contiguous().int()
raw, gam, bet = rn(N, H), rn(H), rn(H)
stats = torch.zeros(N, H // 128, 2, device=DEV); stats[..., 1] = 128.0
vecs = [rn(H) for _ in range(4)]
kf, vf = [(rn(B, S, heads, H) * 0.05).to(dt) for _ in range(Ly)], [rn(B, S, heads, H).to(dt) for _ in range(Ly)]
c = rn(B, S, heads)
y, st_out = torch.empty(N, H, device=DEV), torch.empty(N, H // 128, 2, device=DEV)
descs = []
for l in range(Ly):
    d = L.AttnFusedDesc()
    d.qkvf, d.kcache, d.vfcache, d.anc = qkvf.data_ptr(), kc[l].data_ptr(), vfc[l].data_ptr(), anc.data_ptr()
    d.x_ln.raw, d.x_ln.stats, d.x_ln.gamma, d.x_ln.beta = raw.data_ptr(), stats.data_ptr(), gam.data_ptr(), bet.data_ptr()
    d.x_ln.ld, d.x_ln.parts, d.x_ln.width = H, H // 128, H
    d.self_b, d.ln1_g, d.ln1_b, d.bo = (v.data_ptr() for v in vecs)
    d.kf, d.vf, d.c = kf[l].data_ptr(), vf[l].data_ptr(), c.data_ptr()
    d.y, d.stats_out = y.data_ptr(), st_out.data_ptr()
    d.dtype, d.N, d.H, d.heads, d.S, d.rows_per_image, d.t_cap = L.BF16, N, H, heads, S, R, t_cap
    descs.append(d)
for t in (0, 8, 15, 30):
    for d in descs:
        d.t = t
    for i in range(6):
        L.check(L.lib.mic_attn_fused(C.byref(descs[i % Ly]), L.stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60):
        L.check(L.lib.mic_attn_fused(C.byref(descs[i % Ly]), L.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    print(f"R={R} heads={heads} t={t:2d}: {e0.elapsed_time(e1) / 60 * 1e3:7.1f} us per call", flush=True)
