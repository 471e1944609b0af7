"""Encoder trunk throughput against batch size: do sub-batches small enough for the activations of the early
stages to stay in the memory-side cache run faster per image than one large batch?"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("image-captioning_amd")
W = importlib.import_module("image-captioning_amd.weights")
E = importlib.import_module("image-captioning_amd.encoders")


def main():
    dev = torch.device("cuda:0")
    sd = W.random_resnet50_state(seed=0)
    sd.update(W.random_head_state("transformer", 512, seed=1))
    enc = E.ResNet50Encoder(sd, "transformer", torch.bfloat16, dev)
    total = 256
    for b in (256, 128, 64, 32, 16):
        x = torch.randn(b, 3, 224, 224, device=dev)
        for _ in range(2):
            enc.trunk(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5 * total // b
        for _ in range(reps):
            enc.trunk(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"batch {b:4d}: {dt * 1e3:7.3f} ms per call, {dt * 1e3 * total / b:7.3f} ms per {total} images", flush=True)


if __name__ == "__main__":
    main()
