#!/usr/bin/env python3
"""The benchmark configuration (ResNet-50 + 3-layer transformer, batch 256, beam 3, bf16) against the fp32 CPU oracle on a
larger sample than the test suite affords, the three measures of tests/test_gpu_fullsize.py: relative L2 error of the encoder
features; the reported score of each caption against the oracle's fp32 teacher-forced score of the same tokens; that score
against the oracle's own best beam.  (With random-init weights the step distributions are nearly flat, so which of several
near-tied tokens wins is not a measure; the arithmetic along a sequence is.)
usage: python tests/samples/bf16_parity_sample.py [images]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import image_captioning_amd as amd  # noqa: E402
import bench  # noqa: E402
from oracle import captioner_oracle as O  # noqa: E402  (the checker)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    wl = bench.WORKLOADS["transformer"]
    sd = bench.build_state("transformer")
    model = amd.transformer.ImageCaptioner(amd.Vocabulary.synthetic(bench.V), bench.HEADS, torch.bfloat16, encoder="resnet50",
                                           sync_every=bench.MAX_LEN).load_state_dict(sd).to("cuda:0")
    images = torch.randn(wl["batch"], 3, bench.IMG, bench.IMG, generator=torch.Generator().manual_seed(99)).to("cuda:0")
    feats_gpu = model.encode(images)
    tokens, lengths, score, _ = model.beam_search_ids(feats_gpu, bench.MAX_LEN, wl["beam"])
    print(f"schedule {model.last_schedule}", flush=True)
    t0 = time.time()
    d_self, d_best, same = [], [], 0
    with torch.no_grad():
        for lo in range(0, n, 8):
            fb = O.encoder_transformer(sd, images[lo:lo + 8].cpu())
            ref_tok, best, _ = O.transformer_generate(sd, bench.HEADS, fb, bench.MAX_LEN, None, wl["beam"])
            img = O.tokens_major(fb)
            if lo == 0:
                got = feats_gpu[:8].float().cpu()
                print(f"features, first 8 images: relative L2 error {float((got - img).norm() / img.norm()):.2e}", flush=True)
            for b in range(img.shape[0]):
                ids = tokens[lo + b, : int(lengths[lo + b])].cpu().long()[None]
                logits, _ = O.transformer_decode(sd, bench.HEADS, img[b:b + 1], ids[:, :-1])
                rescored = float(F.log_softmax(logits[0], dim=-1).gather(1, ids[0, 1:, None]).sum())
                d_self.append(abs(float(score[lo + b]) - rescored) / abs(rescored))
                d_best.append((rescored - float(best[b])) / abs(float(best[b])))
                same += ids[0].tolist() == list(ref_tok[b])
            print(f"{lo + 8} images, {time.time() - t0:.0f} s", flush=True)
    ds, db = torch.tensor(d_self), torch.tensor(d_best)
    print(f"{n} captions: reported score against the oracle's score of the same tokens: max {ds.max():.2e}, mean {ds.mean():.2e} (relative)")
    print(f"that score against the oracle's best beam: worst {db.min():+.2e}, best {db.max():+.2e}, mean {db.mean():+.2e} (relative; negative = the oracle's beam is better)")
    print(f"token sequences identical to the oracle's best beam: {same} of {n}")


if __name__ == "__main__":
    main()
