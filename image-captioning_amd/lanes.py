"""Concurrent generate lanes on one GPU.

One generate call is a chain of short dependent launches (the decode steps) behind a throughput-bound encoder, so a
single call leaves the GPU partly idle while it waits on its own dependencies.  Batches are independent
(models/transformer.py:585, models/image_captioner.py:118), so several whole batches can be in flight at once: each
lane is one host thread with its own HIP stream and its own workspaces, all lanes share the device weights, and the
hardware interleaves the lanes' kernels.  Results are bit-identical to running the batches one after another.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Sequence

import torch


class Lanes:
    """`Lanes(model, n)` then `lanes.map(fn, items)`: fn(lane_model, item) is called for every item, at most n at a time,
    lane i handling items i, i+n, i+2n, ...; returns the results in item order.  Every lane's stream has been
    synchronised when map returns, so the results can be used from any stream.  `map(..., lanes=k)` uses the first k
    lanes only.  Keep the number of lanes at or below the process's hardware queues (GPU_MAX_HW_QUEUES, 4 by default):
    streams beyond that share a queue and their lanes take turns instead of overlapping."""

    def __init__(self, model, lanes: int = 3):
        if lanes < 1:
            raise ValueError(f"lanes must be at least 1, got {lanes}")
        if model.device is None:
            raise RuntimeError("move the model to its device before creating lanes")
        self.device = model.device
        self.models = [model] + [model.lane_copy() for _ in range(lanes - 1)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]

    def __len__(self):
        return len(self.models)

    def map(self, fn: Callable, items: Sequence, lanes: int = 0) -> List:
        items = list(items)
        out: List = [None] * len(items)
        errors: List = []
        n = min(len(self.models), lanes) if lanes > 0 else len(self.models)
        for m in self.models:  # a model whose decode step can be one GPU-filling launch uses it only when alone on the GPU
            if hasattr(m, "_concurrent"):
                m._concurrent = n > 1 and len(items) > 1

        def work(lane: int):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[lane]):
                    for i in range(lane, len(items), n):
                        out[i] = fn(self.models[lane], items[i])
                    self.streams[lane].synchronize()
            except BaseException as e:  # re-raised on the calling thread
                errors.append(e)

        main = torch.cuda.current_stream(self.device)
        for s in self.streams[:n]:
            s.wait_stream(main)  # inputs prepared on the caller's stream are complete before a lane reads them
        if n == 1 or len(items) <= 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(lane,)) for lane in range(min(n, len(items)))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        for m in self.models:  # every lane's stream is idle again: a call outside map has the GPU to itself
            if hasattr(m, "_concurrent"):
                m._concurrent = False
        if errors:
            raise errors[0]
        return out
