#!/usr/bin/env python3
"""fp32 LSTM searches (intermediate and basic decoders: E = H = 512, 2 layers, V = 3500, max_len = 30) against the fp32 CPU
oracle on a larger sample than the test suite affords: greedy and beam-3 captions must be identical caption by caption
(models/image_captioner.py:55-159 downstream of the encoder).
usage: python tests/samples/fp32_parity_lstm.py [images]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import image_captioning_amd as amd  # noqa: E402
from oracle import captioner_oracle as O  # noqa: E402  (the checker)
from persist_soak import IdVocab  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    V, E, T = 3500, 512, 30
    vocab = IdVocab(V)
    bad = 0
    for kind, cls in (("intermediate", amd.lstm.IntermediateImageCaptioner), ("basic", amd.lstm.BasicImageCaptioner)):
        sd = amd.weights.random_lstm_state(kind, V, E, E, 2, seed=11)
        model = cls(vocab, torch.float32, encoder="none").load_state_dict(sd).to("cuda:0")
        feats = torch.randn(B, E, generator=torch.Generator().manual_seed(6)).relu()
        for beam in (1, 3):
            t0 = time.time()
            ref = O.lstm_generate(sd, kind, feats, T, None, beam)
            caps = model.generate_from_features(feats, vocab, T, None, beam)
            caps = caps[0] if isinstance(caps, tuple) else caps
            same = sum(c == vocab.encode_as_words(t) for c, t in zip(caps, ref))
            print(f"{kind}, beam {beam}: {same} of {B} captions identical to the oracle's, {time.time() - t0:.0f} s", flush=True)
            bad += same != B
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
