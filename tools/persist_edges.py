#!/usr/bin/env python3
"""Whole searches through the one-launch decode step against the launch-per-sub-layer schedule at the edges of what the
step takes: one image, a ragged last cluster (255, 129 images), 47 generated positions (48 cached), two beams, greedy
decoding (one row per image).  Ids, lengths and scores must be identical and the device error word clear."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import image_captioning_amd as amd  # noqa: E402
from persist_soak import IdVocab  # noqa: E402

DEV = "cuda:0"


def main():
    V, H, Ly, PT, S = 3500, 512, 3, 49, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=5)
    models = {p: amd.transformer.ImageCaptioner(IdVocab(V), 2, torch.bfloat16, encoder="none", persistent=p).load_state_dict(sd).to(DEV)
              for p in (False, True)}
    g = torch.Generator().manual_seed(2)
    bad = 0
    for B, T, K in [(1, 30, 3), (255, 30, 3), (256, 47, 3), (17, 12, 2), (3, 30, 1), (256, 30, 2), (129, 40, 3)]:
        feats = torch.randn(B, S, H, generator=g).relu()
        out = {}
        for p, m in models.items():
            r = m.beam_search_ids(feats, T, K)[:3] if K > 1 else m.sample_ids(feats, T, None)[:2]
            m.check_errors()
            out[p] = [x.cpu() for x in r] + [m.last_schedule]
        same = all(torch.equal(a, b) for a, b in zip(out[False][:-1], out[True][:-1]))
        print(f"{B:4d} images, {T} steps, {K} beam(s): {out[False][-1]} against {out[True][-1]}: {'identical' if same else 'DIFFERENT'}")
        bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
