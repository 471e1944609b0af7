"""Localises a difference between mic_tr_step and mic_tr_step_persistent: one step, every intermediate buffer compared."""
import ctypes as C
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

    def encode_as_words(self, ids):
        return [int(i) for i in ids]


def run(dtype, H, heads, B, R, Ly, T=3):
    V, PT, S = 500, 16, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=5)
    model = amd.transformer.ImageCaptioner(IdVocab(V), heads, dtype, encoder="none").load_state_dict(sd).to(DEV)
    g = torch.Generator().manual_seed(23)
    N = B * R
    feats = torch.randn(B, S, H, generator=g).relu().to(DEV, dtype).contiguous()
    caps = torch.randint(4, V, (N, T), generator=g, dtype=torch.int32).to(DEV)
    ws, t = model._workspace(N, T, S, R, max(R, 1), 0)
    res = {}
    for fn_name in ("mic_tr_step", "mic_tr_step_persistent"):
        for k in ("kc", "vc", "vfc", "y", "stats", "qkvf", "ffh", "logits"):
            t[k].zero_()
        L.check(L.lib.mic_tr_cross_kv(C.byref(model.dec), C.byref(ws), feats.data_ptr(), L.stream_ptr()), "cross_kv")
        fn = getattr(L.lib, fn_name)
        snaps = []
        for p in range(T):
            L.check(fn(C.byref(model.dec), C.byref(ws), caps.data_ptr(), T, None, p, 0, L.stream_ptr()), fn_name)
            torch.cuda.synchronize()
            snaps.append({k: t[k].clone() for k in ("qkvf", "y", "stats", "ffh", "logits", "kc", "vfc", "x", "xt")})
        res[fn_name] = snaps
        print(fn_name, "err word", int(t["sync"][-64].item()))
    for p in range(T):
        for k in ("x", "xt", "qkvf", "y", "stats", "ffh", "logits", "kc", "vfc"):
            a, b = res["mic_tr_step"][p][k].float(), res["mic_tr_step_persistent"][p][k].float()
            d = (a - b).abs()
            bad = (d > 0).nonzero()
            msg = "equal" if bad.numel() == 0 else f"DIFF n={bad.shape[0]} max={d.max().item():.4g} first={bad[0].tolist()} last={bad[-1].tolist()}"
            print(f"  dtype={dtype} B={B} R={R} L={Ly} step {p} {k:7s} {tuple(a.shape)} {msg}")


if __name__ == "__main__":
    run(torch.bfloat16, 512, 2, 16, 1, 1)
    run(torch.bfloat16, 512, 2, 16, 3, 1)
    run(torch.bfloat16, 512, 2, 256, 3, 3, T=2)
