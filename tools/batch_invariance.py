#!/usr/bin/env python3
"""Does the caption of an image depend on the batch it is in?  Encodes and captions 64 images once as a batch of 64 and once as
the first 64 of a batch of 256 (bf16, the benchmark's random-init model, whose near-uniform logits make beam search as
sensitive to a last-bit difference as it gets).  The only batch-dependent arithmetic is the in-launch split of K, which the
stage-3 3x3 layers take at batch 256 and not at batch 64 (fp32 partial sums regrouped: a bf16 output can move by one ulp);
MIC_IGEMM8_SPLITK=0 switches it off and the two runs then agree bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_captioning_amd as amd  # noqa: E402
import bench  # noqa: E402


def main():
    dev = "cuda:0"
    sd = bench.build_state("transformer")
    m = amd.transformer.ImageCaptioner(amd.Vocabulary.synthetic(bench.V), bench.HEADS, torch.bfloat16, encoder="resnet50",
                                       sync_every=30).load_state_dict(sd).to(dev)
    img = torch.randn(256, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(5), device=dev)
    f256 = m.encode(img).float().clone()
    f64 = m.encode(img[:64].contiguous()).float().clone()
    d = (f256[:64] - f64).abs()
    print(f"split K {'off' if os.environ.get('MIC_IGEMM8_SPLITK') == '0' else 'on'}: features of the same 64 images in a batch of 256 and of 64: "
          f"max |d| {d.max().item():.3g} (largest feature {f64.abs().max().item():.3g}), {100 * (d == 0).float().mean().item():.2f} % of the elements equal")
    t256 = m.beam_search_ids(m.encode(img), 30, 3)[0][:64].cpu()
    t64 = m.beam_search_ids(m.encode(img[:64].contiguous()), 30, 3)[0].cpu()
    print(f"captions identical: {int((t256 == t64).all(dim=1).sum())} of 64")


if __name__ == "__main__":
    main()
