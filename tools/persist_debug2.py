"""Stops the one-launch step after a seam and checks what it left against torch on the host."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
import image_captioning_amd as amd  # noqa: E402

L = amd._lib
DEV = "cuda:0"


class IdVocab:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def str_to_idx(self, s):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[s]


def main():
    stop = int(os.environ.get("MIC_PERSIST_STOP", "-1"))
    dtype, H, heads, B, R, Ly, T = torch.bfloat16, 512, 2, 16, 1, 1, 1
    V, PT, S = 500, 16, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=5)
    model = amd.transformer.ImageCaptioner(IdVocab(V), heads, dtype, encoder="none").load_state_dict(sd).to(DEV)
    g = torch.Generator().manual_seed(23)
    N = B * R
    feats = torch.randn(B, S, H, generator=g).relu().to(DEV, dtype).contiguous()
    caps = torch.randint(4, V, (N, T), generator=g, dtype=torch.int32).to(DEV)
    ws, t = model._workspace(N, T, S, R, max(R, 1), 0)
    for k in ("kc", "vc", "vfc", "y", "stats", "qkvf", "ffh", "logits"):
        t[k].zero_()
    L.check(L.lib.mic_tr_cross_kv(C.byref(model.dec), C.byref(ws), feats.data_ptr(), L.stream_ptr()), "cross_kv")
    L.check(L.lib.mic_tr_step_persistent(C.byref(model.dec), C.byref(ws), caps.data_ptr(), T, None, 0, 0, L.stream_ptr()), "persist")
    torch.cuda.synchronize()
    print("stop", stop, "err", int(t["sync"][-64].item()), "counters", t["sync"][:8].tolist())
    y = t["y"].float().cpu()
    st = t["stats"].cpu()
    # statistics of y per 128 columns
    yp = y.view(N, 4, 128)
    mean = yp.mean(-1)
    m2 = ((yp - mean[..., None]) ** 2).sum(-1)
    for i in (0, 1):
        print(f"stats[{i}] vs y: mean diff {(st[i, :, :, 0] - mean).abs().max().item():.4g}  m2 rel diff {((st[i, :, :, 1] - m2).abs() / (m2.abs() + 1e-6)).max().item():.4g}")
    g2 = sd["decoder_layers.0.cross_attention.layer_norm.weight"].float()
    b2 = sd["decoder_layers.0.cross_attention.layer_norm.bias"].float()
    x = torch.nn.functional.layer_norm(y, (H,), g2, b2, 1e-5)
    w1 = sd["decoder_layers.0.ff.seq.0.weight"].float()
    b1 = sd["decoder_layers.0.ff.seq.0.bias"].float()
    ff = torch.relu(x.to(dtype).float() @ w1.to(dtype).float().t() + b1)
    got = t["ffh"].float().cpu()
    print("ffh vs torch(relu(LN2(y) W1 + b1)):", (got - ff).abs().max().item(), "ref max", ff.abs().max().item(), "got max", got.abs().max().item())
    print("y[0,:8]", y[0, :8].tolist())
    print("dbg rs_x / g / b / xs:", st[1].reshape(-1)[:128].view(16, 8)[:4, :5].tolist())
    print("expected mean/rstd row0-3:", y.mean(-1)[:4].tolist(), (1.0 / (y.var(-1, unbiased=False) + 1e-5).sqrt())[:4].tolist())
    print("g2[:4]", g2[:4].tolist(), "b2[:4]", b2[:4].tolist(), "x exp row0[:4]", x[0, :4].tolist())
    print("ffh got[0,:8]", got[0, :8].tolist(), "exp", ff[0, :8].tolist())


main()
