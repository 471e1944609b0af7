#!/usr/bin/env python3
"""Soak run of the one-launch decode step: many whole beam searches at the benchmark's shape (256 images, beam 3, H = 512,
3 layers, V = 3500, 30 steps), fresh features every search, each compared id for id, length for length and score bit for
bit with the launch-per-sublayer schedule, and the device error word checked after every search.  A lost hand-off between
workgroups shows up as a differing search or a non-zero error word, never as a hang (every wait in the kernel is bounded).
usage: persist_soak.py [searches [steps]]   (more than 31 steps take the six-positions-per-wave build of the attention phase)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

DEV = "cuda:0"


class IdVocab:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def str_to_idx(self, s):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[s]

    def encode_as_words(self, ids):
        return [str(i) for i in ids]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    V, H, Ly, PT, S, B = 3500, 512, 3, T + 1, 49, 256
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=3)
    models = {p: amd.transformer.ImageCaptioner(IdVocab(V), 2, torch.bfloat16, encoder="none", persistent=p).load_state_dict(sd).to(DEV)
              for p in (False, True)}
    g = torch.Generator().manual_seed(11)
    bad = steps = 0
    t0 = time.time()
    for i in range(n):
        feats = torch.randn(B, S, H, generator=g).relu()
        out = {}
        for p, m in models.items():
            tok, ln, sc = m.beam_search_ids(feats, T, 3)[:3]
            m.check_errors()
            out[p] = (tok.cpu(), ln.cpu(), sc.cpu())
            assert m.last_schedule == ("persistent" if p else "fused14"), m.last_schedule
        same = all(torch.equal(a, b) for a, b in zip(out[False], out[True]))
        bad += not same
        steps += T - 1
        if (i + 1) % 50 == 0 or not same:
            print(f"search {i + 1}/{n}: {steps} one-launch steps so far, {bad} differing searches, {time.time() - t0:.0f} s", flush=True)
    print(f"{n} searches of {B} images, {steps} one-launch steps ({steps * 13} hand-offs per cluster): {bad} differing searches, error word clear after each")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
