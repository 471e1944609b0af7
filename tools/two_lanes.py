#!/usr/bin/env python3
"""Development aid: throughput of L concurrent generate lanes in one process (one host thread, one stream, one
model instance with its own workspaces per lane; each lane captions whole batches of 256) against one lane.

    python tools/two_lanes.py [lanes]
"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import image_captioning_amd as amd  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = 8
sd = bench.build_state()
vocab = amd.Vocabulary.synthetic(bench.V)
models = [amd.transformer.ImageCaptioner(vocab, bench.HEADS, torch.bfloat16, sync_every=bench.MAX_LEN).load_state_dict(sd).to(dev)
          for _ in range(LANES)]
images = [torch.randn(bench.BATCH, 3, 224, 224, device=dev) for _ in range(LANES)]
streams = [torch.cuda.Stream() for _ in range(LANES)]


def lane(i, steps):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[i]):
        for _ in range(steps):
            feats = models[i].encode(images[i])
            models[i].beam_search_ids(feats, bench.MAX_LEN, bench.BEAM)
        streams[i].synchronize()


def run(lanes, steps):
    th = [threading.Thread(target=lane, args=(i, steps)) for i in range(lanes)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for lanes in (1, LANES, 1, LANES):
    run(lanes, 2)
    dt = run(lanes, K)
    print(f"{lanes} lane(s): {dt / (K * lanes) * 1e3:7.3f} ms per batch, {bench.BATCH * K * lanes / dt:9.1f} images/s", flush=True)
