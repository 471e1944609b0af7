#!/usr/bin/env python3
"""Encoder of one batch of 256 against the two halves of the batch on two host threads and streams (development aid)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import image_captioning_amd as amd  # noqa: E402

DEV = "cuda:0"
sd = bench.build_state("transformer")
model = amd.transformer.ImageCaptioner(amd.Vocabulary.synthetic(bench.V), bench.HEADS, torch.bfloat16).load_state_dict(sd).to(DEV)
m2 = model.lane_copy()
images = torch.randn(256, 3, 224, 224, device=DEV)
halves = (images[:128].contiguous(), images[128:].contiguous())
streams = (torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV))


def whole(n):
    for _ in range(n):
        model.encode(images)


def split(n):
    def work(m, x, s):
        torch.cuda.set_device(DEV)
        with torch.cuda.stream(s):
            for _ in range(n):
                m.encode(x)
    th = [threading.Thread(target=work, args=(m, x, s)) for m, x, s in zip((model, m2), halves, streams)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for fn, nm in ((whole, "one batch of 256"), (split, "two halves of 128 on two threads / streams")):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(10)
    torch.cuda.synchronize()
    print(f"{nm}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 256 images")
