"""World-size-2 gloo run of the data-parallel plumbing: contiguous sharding and the final gather of
token ids / lengths / scores, including a ragged last shard and the gather-to-rank-0 form."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from image_captioning_amd import distributed as D
    r, _, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(n_total, rank, world)
    T = 6
    idx = torch.arange(lo, hi)
    tokens = (idx[:, None] * 10 + torch.arange(T)[None]).to(torch.int32)   # "captions" derived from the global image index
    lengths = (idx % T + 1).to(torch.int32)
    scores = -idx.float()
    out_all = D.gather_captions(tokens, lengths, scores, n_total)
    out_dst = D.gather_captions(tokens, lengths, scores, n_total, dst=0)
    # the whole job: micro-batches of 3 over this rank's slice, one gather at the end; then a job with fewer items than ranks
    class Host:
        device = None

    def caption(_model, ix):
        return (ix[:, None] * 10 + torch.arange(T)[None]).to(torch.int32), (ix % T + 1).to(torch.int32), -ix.float()

    job = D.caption_dataset(Host(), caption, lambda a, b: torch.arange(a, b), n_total, T, batch=3)
    assert all(torch.equal(x, y) for x, y in zip(job, out_all))
    tiny = D.caption_dataset(Host(), caption, lambda a, b: torch.arange(a, b), 1, T, batch=3)
    assert tiny[0].tolist() == [[t for t in range(T)]] and tiny[1].tolist() == [1]
    # a rank that fails inside its shard must not leave the others blocked in the gather: every rank raises
    def broken(_model, ix):
        if rank == 1:
            raise ValueError("rank 1 cannot caption")
        return caption(_model, ix)

    try:
        D.caption_dataset(Host(), broken, lambda a, b: torch.arange(a, b), n_total, T, batch=3)
        raise AssertionError("caption_dataset returned although rank 1 failed")
    except ValueError as e:
        assert rank == 1 and "cannot caption" in str(e)
    except RuntimeError as e:
        assert rank == 0 and "another rank failed" in str(e)
    q.put((rank, [t.tolist() for t in out_all], None if out_dst[0] is None else [t.tolist() for t in out_dst]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, n_total = 2, 7  # ragged: rank 0 gets 4 images, rank 1 gets 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = 6
    want_tok = [[i * 10 + t for t in range(T)] for i in range(n_total)]
    want_len = [i % T + 1 for i in range(n_total)]
    want_sc = [-float(i) for i in range(n_total)]
    for rank, allr, dstr in res:
        assert allr == [want_tok, want_len, want_sc]      # identical on every rank, in global image order
        if rank == 0:
            assert dstr == [want_tok, want_len, want_sc]
        else:
            assert dstr is None


def test_single_process_gather_is_identity():
    sys.path.insert(0, ROOT)
    from image_captioning_amd import distributed as D
    t, l, s = torch.zeros(3, 4, dtype=torch.int32), torch.ones(3, dtype=torch.int32), torch.zeros(3)
    a, b, c = D.gather_captions(t, l, s, 3)
    assert a.shape == (3, 4) and b.tolist() == [1, 1, 1] and c.shape == (3,)
