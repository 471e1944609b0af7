#!/usr/bin/env python3
"""Prints a checksum over the ids and lengths of many LSTM beam searches (intermediate model, 64 images, beam 3, bf16 and
fp32) on fixed seeds.  Run once as is and once with MIC_NO_CELL_FUSE=1: the gate GEMM with the cell in its epilogue must
return the captions of the two-launch form.
usage: lstm_cell_check.py [searches]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402

DEV = "cuda:0"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    V, E, T = 3500, 512, 30
    h = hashlib.sha256()
    keep = []
    for dtype in (torch.bfloat16, torch.float32):
        sd = amd.weights.random_lstm_state("intermediate", V, E, E, 2, seed=7)
        model = amd.lstm.IntermediateImageCaptioner(amd.Vocabulary.synthetic(V), dtype, encoder="none").load_state_dict(sd).to(DEV)
        g = torch.Generator().manual_seed(3)
        for _ in range(n):
            feats = torch.randn(64, E, generator=g).to(DEV)
            tok, ln = model.beam_search_ids(feats, T, 3)[:2]
            h.update(tok.cpu().numpy().tobytes())
            h.update(ln.cpu().numpy().tobytes())
            keep.append(tok.cpu())
    if len(sys.argv) > 2:  # a file to keep the ids in, for a caption-by-caption comparison of two runs
        torch.save(torch.stack(keep), sys.argv[2])
    print(f"{n} searches per dtype, fused={'MIC_NO_CELL_FUSE' not in os.environ}: {h.hexdigest()[:24]}")


if __name__ == "__main__":
    main()
