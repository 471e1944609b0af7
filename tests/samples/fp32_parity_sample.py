#!/usr/bin/env python3
"""fp32 search at the benchmark's model size (H = 512, 2 heads, 3 layers, V = 3500, S = 49, max_len = 30) against the fp32 CPU
oracle on a larger sample than the test suite affords: token ids must be identical caption by caption, summed
log-probabilities within 1e-3 (the reference's algorithm: models/transformer.py:458-640).  Prints one line per search mode.
With a hidden size of 256 (rows of 1 KiB in fp32) and at most 256 images the search takes the one-launch step, the schedule
the bf16 benchmark times; at 512 the launch-per-sub-layer schedules.
usage: python tests/samples/fp32_parity_sample.py [images [hidden size]]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import image_captioning_amd as amd  # noqa: E402
from oracle import captioner_oracle as O  # noqa: E402  (the checker)
from persist_soak import IdVocab  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    V, Ly, PT, S, heads, T = 3500, 3, 40, 49, 2, 30
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=21)
    g = torch.Generator().manual_seed(9)
    for k in list(sd):  # biases and LayerNorm parameters away from their trivial initial values
        if k.endswith("in_proj_bias") or k.endswith("out_proj.bias") or k.endswith("layer_norm.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
    vocab = IdVocab(V)
    model = amd.transformer.ImageCaptioner.from_state_dict(sd, vocab, heads, torch.float32, "cuda:0")
    feats_bchw = torch.randn(B, H, 7, 7, generator=g).relu()
    feats = O.tokens_major(feats_bchw)
    bad = 0
    for beam in (1, 3):
        t0 = time.time()
        ref_tok, ref_lp, _ = O.transformer_generate(sd, heads, feats_bchw, T, None, beam)
        caps, lp = model.generate_from_features(feats, vocab, T, None, beam)
        same = sum(c == vocab.encode_as_words(t) for c, t in zip(caps, ref_tok))
        dlp = float(np.abs(lp.cpu().numpy() - ref_lp.numpy()).max())
        print(f"H = {H}, beam {beam}: {same} of {B} captions identical to the oracle's, max |d log-prob| {dlp:.2e}, schedule {model.last_schedule}, "
              f"{time.time() - t0:.0f} s", flush=True)
        bad += same != B or dlp > 1e-3
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
