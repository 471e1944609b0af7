"""Data-parallel captioning: images are independent (the reference's beam search already handles
one image at a time, models/transformer.py:585), so ranks take contiguous slices of the image set,
run the whole generate path locally with replicated weights, and exchange nothing until one final
gather of token ids and scores.  One process per GPU; `torch.distributed` backend "nccl" is RCCL
over xGMI on ROCm, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`: ceil(n/world) items per rank, the tail ranks short or empty."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_captions(tokens: torch.Tensor, lengths: torch.Tensor, scores: torch.Tensor, n_total: int,
                    dst: Optional[int] = None):
    """Final exchange: every rank contributes its shard's int32 token matrix (n_local, T), lengths and
    fp32 scores; returns the concatenation over ranks in rank order, trimmed to n_total rows
    (on every rank, or only on `dst` when given; other ranks get None).  Shards are padded to the
    per-rank maximum and packed so that one fixed-size collective carries everything."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tokens[:n_total], lengths[:n_total], scores[:n_total]
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    T = tokens.shape[1]
    # one message per rank: [per][T + 2] int32 = token ids | length | score bits, so the whole exchange
    # is a single collective (its latency, not its 4*(T+2) bytes per image, is what a step pays)
    pack = torch.empty(per, T + 2, dtype=torch.int32, device=tokens.device)
    n = tokens.shape[0]
    pack[:n, :T] = tokens
    pack[:n, T] = lengths
    pack[:n, T + 1] = scores.to(torch.float32).contiguous().view(torch.int32)
    if n < per:
        pack[n:, :T] = -1
        pack[n:, T:] = 0
    if dst is None:
        buf = torch.empty(world * per, T + 2, dtype=torch.int32, device=pack.device)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(buf, pack)
        else:  # gloo gathers host tensors
            host = pack.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            buf = torch.cat(parts, dim=0).to(pack.device)
    else:
        src = pack if dist.get_backend() == "nccl" else pack.cpu()
        parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
        dist.gather(src, parts, dst=dst)
        if rank != dst:
            return None, None, None
        buf = torch.cat(parts, dim=0).to(pack.device)
    buf = buf[:n_total]
    return buf[:, :T].contiguous(), buf[:, T].contiguous(), buf[:, T + 1].contiguous().view(torch.float32)


def _agree_ok(error: Optional[BaseException], device) -> None:
    """Ranks agree on success before the one collective of the job: a rank that failed (or got the wrong width from
    caption()) would otherwise leave while the others block in the gather.  One int all-reduce; every rank raises
    when any rank failed."""
    bad = 0 if error is None else 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        flag = torch.tensor([bad], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad_any = int(flag.item())
    else:
        bad_any = bad
    if error is not None:
        raise error
    if bad_any:
        raise RuntimeError("caption_dataset: another rank failed before the gather; this rank's shard is discarded")


def caption_dataset(model, caption, make_batch, n_total: int, n_tokens: int, batch: int = 256, lanes=1):
    """Caption items 0..n_total-1 data-parallel: this rank takes its contiguous slice (shard_range), walks it in
    micro-batches of `batch`, and every rank returns the gathered (tokens int32 (n_total, n_tokens), lengths, scores)
    in item order.  `make_batch(lo, hi)` returns the input of items [lo, hi) on the model's device;
    `caption(model, input)` returns (tokens (n, n_tokens) int32, lengths (n,), scores (n,)).  `lanes` is a number of
    micro-batches to keep in flight on the GPU (a lanes.Lanes pool is created for the call) or an existing
    lanes.Lanes pool, which keeps its streams, workspaces and captured graphs from call to call; the result does
    not depend on it.  The only collectives are a one-int agreement that every rank got through its shard and
    the final gather; when any rank failed, every rank raises instead of entering the gather."""
    if n_total < 0 or batch < 1 or n_tokens < 1:
        raise ValueError(f"bad sizes n_total={n_total} batch={batch} n_tokens={n_tokens}")
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    lo, hi = shard_range(n_total, rank, world)
    spans = [(a, min(hi, a + batch)) for a in range(lo, hi, batch)]
    dev = getattr(model, "device", None) or "cpu"

    def work(m, span):
        tokens, lengths, scores = caption(m, make_batch(*span))
        return tokens.clone(), lengths.clone(), scores.clone()  # a model may hand out views of buffers it reuses

    error: Optional[BaseException] = None
    tokens = lengths = scores = None
    try:
        pool = None if isinstance(lanes, int) else lanes
        if pool is None and lanes > 1 and len(spans) > 1:
            from .lanes import Lanes
            pool = Lanes(model, lanes)
        if pool is not None and len(pool) > 1 and len(spans) > 1:
            parts = pool.map(work, spans)
        else:
            parts = [work(model, s) for s in spans]
        if parts:
            tokens = torch.cat([p[0] for p in parts]).to(torch.int32)
            lengths = torch.cat([p[1] for p in parts]).to(torch.int32)
            scores = torch.cat([p[2] for p in parts]).to(torch.float32)
            if tokens.shape[1] != n_tokens:
                raise ValueError(f"caption() returned {tokens.shape[1]} tokens per item, expected {n_tokens}")
        else:  # more ranks than micro-batches: an empty shard still takes part in the gather
            tokens = torch.empty(0, n_tokens, dtype=torch.int32, device=dev)
            lengths = torch.empty(0, dtype=torch.int32, device=dev)
            scores = torch.empty(0, dtype=torch.float32, device=dev)
    except Exception as e:  # noqa: BLE001 - re-raised on every rank after the agreement
        error = e
    _agree_ok(error, dev)
    return gather_captions(tokens, lengths, scores, n_total)
