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
