#!/usr/bin/env python3
"""Headline benchmark: images captioned / second, transformer captioner (ResNet-50 encoder +
3-layer transformer decoder, H=512, 2 heads, V=3500), beam=3, max_len=30, 224x224 synthetic
images, batch 256 per GPU, bf16 weights/GEMM inputs with fp32 accumulation.

    python bench.py --gpus N --steps K --warmup W

One "step" = one whole generate call (encoder + cross-K/V + 30 beam-search steps + best-beam
extraction) over one batch of 256 images that already sit in HBM, followed by the gather of token
ids across ranks.  String detokenisation is excluded (the metric is defined on token ids).
The timed region runs the batches one after another, so that the per-kernel durations behind the
roofline figures are those of kernels that have the GPU to themselves.  Batches are independent, so a
job can keep several in flight (--lanes N: one host thread, stream and set of workspaces per lane,
image_captioning_amd/lanes.py); the throughput of 3 lanes is measured after the timed region and
reported as "concurrent_lanes".  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH, IMG, MAX_LEN, BEAM = 256, 224, 30, 3
V, H, LAYERS, HEADS, POS_ROWS = 3500, 512, 3, 2, 40
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of one MI355X
MFMA_F32_PEAK_TFLOPS = 157.3    # f32-input MFMA (= the f32 vector rate)


# secondary workloads (parity-test configurations of the same path; the default is the headline one)
WORKLOADS = {
    "transformer": dict(batch=256, beam=3, label="configs[2]: ResNet-50 encoder + 3-layer transformer decoder (H=512, 2 heads, V=3500), "
                        "beam=3, max_len=30, 224x224, batch 256 per GPU, bf16; random-init weights"),
    "swin": dict(batch=512, beam=5, label="configs[3]: Swin-V2-S encoder + 3-layer transformer decoder (H=512, 2 heads, V=3500), beam=5, "
                 "max_len=30, 224x224, batch 512 per GPU, bf16; random-init weights"),
    "swin_t": dict(batch=512, beam=5, label="configs[3] as worded: Swin-V2-T encoder (depths 2,2,6,2; the reference itself builds swin_v2_s, "
                   "which is --workload swin) + 3-layer transformer decoder (H=512, 2 heads, V=3500), beam=5, max_len=30, 224x224, "
                   "batch 512 per GPU, bf16; random-init weights"),
    "lstm": dict(batch=64, beam=3, label="configs[1]: ResNet-50 encoder + 2-layer LSTM decoder (intermediate model, E=H=512, V=3500), "
                 "beam=3, max_len=30, 224x224, batch 64 per GPU, bf16; random-init weights"),
    "basic": dict(batch=8, beam=1, label="configs[0]: basic model (ResNet-50 encoder with avg-pool + Linear, 2-layer LSTM decoder fed the image "
                  "feature as its first time step, E=H=512, V=3500), greedy, max_len=30, 224x224, batch 8, fp32 (the reference's own "
                  "CPU-runnable case: its CPU time is cpu_baseline); random-init weights"),
    "job": dict(batch=256, beam=3, label="configs[4]: transformer model (ResNet-50 encoder + 3-layer transformer decoder, H=512, 2 heads, "
                "V=3500), beam=3, max_len=30, 224x224, bf16, one job of --job-images synthetic images sharded contiguously over the "
                "ranks in micro-batches of 256, no collective until one final gather of the token ids; random-init weights"),
}


def build_state(workload: str = "transformer"):
    import image_captioning_amd as amd
    if workload in ("swin", "swin_t"):
        sd = amd.weights.random_swin_state(0, depths=(2, 2, 6, 2) if workload == "swin_t" else (2, 2, 18, 2))
        sd.update(amd.weights.random_head_state("swin", H, 0, in_dim=768))
    elif workload in ("lstm", "basic"):
        kind = "intermediate" if workload == "lstm" else "basic"
        sd = amd.weights.random_resnet50_state(0)
        sd.update(amd.weights.random_head_state(kind, H, 0))
        sd.update(amd.weights.random_lstm_state(kind, V, H, H, 2, seed=0))
        return sd
    else:
        sd = amd.weights.random_resnet50_state(0)
        sd.update(amd.weights.random_head_state("transformer", H, 0))
    sd.update(amd.weights.random_transformer_state(V, H, LAYERS, POS_ROWS, seed=0))
    return sd


def cpu_baseline(sd, n_images: int, workload: str, beam: int):
    """The oracle (reference-as-written search, fp32 PyTorch-CPU) on the first images of the same
    synthetic workload, timed on this host's cores."""
    import torch
    from oracle import captioner_oracle as O
    # the box's share of host cores for one GPU is 16 (OMP_NUM_THREADS there); os.cpu_count() reports the whole machine
    host_cores = os.cpu_count() or 1
    threads = min(int(os.environ.get("OMP_NUM_THREADS", "16")), host_cores)
    torch.set_num_threads(threads)
    img = torch.randn(n_images, 3, IMG, IMG, generator=torch.Generator().manual_seed(1234))
    t0 = time.perf_counter()
    with torch.no_grad():
        if workload == "lstm":
            O.lstm_generate(sd, "intermediate", O.encoder_intermediate(sd, img), MAX_LEN, None, beam)
        elif workload == "basic":
            O.lstm_generate(sd, "basic", O.encoder_basic(sd, img), MAX_LEN, None, beam)
        else:
            feats = (O.encoder_swin if workload in ("swin", "swin_t") else O.encoder_transformer)(sd, img)
            O.transformer_generate(sd, HEADS, feats, MAX_LEN, None, beam)
    dt = time.perf_counter() - t0
    how = f"per-image beam={beam} search" if beam > 1 else "batched greedy decoding"
    return {"value": round(n_images / dt, 3), "unit": "images/s", "cores": threads, "host_cpu_count": host_cores, "kind": "port",
            "sample": f"{n_images} images of the same workload (encoder + {how} with full-prefix "
                      f"recompute, fp32 torch CPU, {threads} threads of the host's {host_cores} logical CPUs), {dt:.1f} s"}


HBM_PEAK_TBS = 8.0  # HBM3E peak of one MI355X


def decode_roofline(ms: float, batch: int, beam: int, steps: int, S: int):
    """Algorithmic bytes of one beam search at bf16 (2 bytes) over `steps` executed steps, N = batch*beam rows: per step
    the decoder weights once (12 H^2 per layer: both attentions' projections 8, feed-forward 4) plus the vocabulary
    matrix; the cross-attention keys and values of every image and layer (2 S H per image and layer); the self-attention
    cache (read positions 0..t of every row, write one: 2 H each); logits written and read in fp32; and once per search
    the cross-K/V projection (image tokens read, keys/values written).  `traffic_model` is the same count for the layout
    the kernels actually use, in which the query / output projections of both attentions are folded into the keys and
    values (cross keys and values and cached self values `heads` times as wide, 4 H^2 fewer weights per layer)."""
    N, e = batch * beam, 2
    weights = (LAYERS * 12 * H * H + V * H) * e
    cross = LAYERS * 2 * batch * S * H * e
    cache = sum(LAYERS * 2 * N * H * e * (t + 1) for t in range(steps))  # t reads + 1 write at step t
    logits = 2 * N * V * 4
    once = batch * S * H * e + cross
    total = steps * (weights + cross + logits) + cache + once
    w_f = (LAYERS * (4 + 2 * HEADS) * H * H + V * H) * e  # packed q | k | folded values, feed-forward
    cross_f = cross * HEADS
    cache_f = sum(LAYERS * (1 + HEADS) * N * H * e * (t + 1) for t in range(steps))
    folded_total = steps * (w_f + cross_f + logits) + cache_f + batch * S * H * e + cross + cross_f
    tbs = total / (ms * 1e-3) / 1e12
    return {"bound": "hbm", "phase": "cross-K/V + beam steps + best-beam extraction", "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS,
            "unit": "TB/s", "frac": round(tbs / HBM_PEAK_TBS, 4), "ms": round(ms, 3), "algorithmic_bytes": int(total),
            "traffic_model": int(folded_total)}


def lstm_decode_roofline(ms: float, batch: int, beam: int, steps: int, layers: int = 2):
    """Algorithmic bytes of one LSTM beam search at bf16 over `steps` executed steps, N = batch*beam rows: per step the
    packed [W_ih | W_hh] of every layer ((in + H) x 4H, E = H) and the vocabulary matrix once; per row the layer input
    [x | h] read, the four gates written and read in fp32, the cell state read and written in fp32, the new hidden state
    written; logits written and read in fp32."""
    N, e = batch * beam, 2
    weights = (layers * 2 * H * 4 * H + V * H) * e
    rows = layers * N * (2 * H * e + 2 * 4 * H * 4 + 2 * H * 4 + H * e) + 2 * N * V * 4
    total = steps * (weights + rows)
    tbs = total / (ms * 1e-3) / 1e12
    return {"bound": "hbm", "phase": "initial state + beam steps + best-beam extraction", "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS,
            "unit": "TB/s", "frac": round(tbs / HBM_PEAK_TBS, 4), "ms": round(ms, 3), "algorithmic_bytes": int(total)}


def measured_traffic():
    """HBM bytes per launch of the 128-row-tile bf16 igemm kernels on this workload, from the committed
    rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs of this script; a live
    run cannot collect hardware counters on itself).  None when the summary file is absent."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return round(json.load(f)["igemm_bf16_128row_tiles"]["hbm_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def run_job(args, amd, D, model, sd, dev, rank, world, wl):
    """configs[4]: one job of --job-images images, sharded contiguously over the ranks, micro-batches of 256, nothing
    exchanged until the final gather (distributed.caption_dataset).  Image i is a function of i alone (one of 1,024 seeded
    base images plus a small offset), so the result does not depend on the number of ranks; rank 0 checks that on a
    1,024-image prefix against the single-process schedule."""
    import torch
    import torch.distributed as dist
    n, mb, beam, t_cap = args.job_images, args.batch or wl["batch"], wl["beam"], MAX_LEN + 1
    P = min(1024, max(1, n))
    base = torch.randn(P, 3, IMG, IMG, generator=torch.Generator(device=dev).manual_seed(1234), device=dev)

    def make_batch(lo, hi):
        idx = torch.arange(lo, hi, device=dev)
        return base[idx % P] + (idx // P).to(torch.float32).view(-1, 1, 1, 1) * 1e-3

    def caption(m, images):
        tokens, lengths, score, _ = m.beam_search_ids(m.encode(images), MAX_LEN, beam)
        return tokens, lengths, score

    pool = amd.lanes.Lanes(model, max(1, args.lanes))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = min(n, world * mb * max(2, 2 * len(pool)))  # set-up, not measured: workspaces and captured graphs of every lane
    D.caption_dataset(model, caption, make_batch, warm, t_cap, mb, pool)
    fence()
    t0 = time.perf_counter()
    tokens, lengths, score = D.caption_dataset(model, caption, make_batch, n, t_cap, mb, pool)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        # the prefix lies inside rank 0's shard and ends on a micro-batch boundary, so that the recomputation below sees
        # the same micro-batches as the job did (tile choices depend on the batch size; bf16 bits on the tile choice)
        lo, hi = D.shard_range(n, 0, world)
        pre = min(1024, hi - lo)
        pre = pre // mb * mb if pre >= mb else pre
        ref = [caption(model, make_batch(lo, min(pre, lo + mb))) for lo in range(0, pre, mb)]
        same = all(torch.equal(torch.cat([r[i] for r in ref]), x[:pre]) for i, x in enumerate((tokens, lengths, score)))
        steps = (hi - lo + mb - 1) // mb
        weights = torch.arange(1, t_cap + 1, device=dev, dtype=torch.int64)
        res = {
            "metric": f"images captioned/sec (224^2, beam={beam}, max_len=30)", "value": round(n / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": steps, "warmup": 0, "ms_per_step": round(elapsed / max(1, steps) * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["label"], "job_images": n, "micro_batch": mb, "micro_batches_per_rank": steps,
                       "parallelism": f"dp{world}", "lanes_per_gpu": len(pool), "collectives": "one final gather of token ids, lengths, scores",
                       "executed_decode_steps": int(lengths.max().item()) - 1},
            "job_seconds": round(elapsed, 3),
            "prefix_check": {"images": pre, "identical_to_single_process_schedule": bool(same),
                             "token_checksum": int(((tokens[:pre].to(torch.int64) + 1) * weights).sum().item())},
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(res), flush=True)
        if not same:
            raise SystemExit("bench.py: the gathered ids of the first images differ from the single-process schedule")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=1, help="whole batches in flight per GPU in the timed region (host threads with a stream "
                    "and workspaces each); with 1, the throughput of 3 lanes is measured after the timed region and reported beside it")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--decode-groups", type=int, default=1, help="concurrent beam-search sub-batches per GPU")
    ap.add_argument("--concurrent-steps", type=int, default=240, help="steps of the 3-lane throughput measurement that follows the timed "
                    "region at --lanes 1: about two seconds of GPU work, long enough for an outside utilisation sampler to see the GPU "
                    "busy (0 skips it, e.g. under a profiler, so that the kernel statistics are those of the timed region's schedule)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="transformer")
    ap.add_argument("--cpu-images", type=int, default=128, help="sample size of the CPU baseline (0 disables it)")
    ap.add_argument("--job-images", type=int, default=100_000, help="--workload job: images of the whole job, all ranks together")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start one rank per GPU as child processes and pass their exit code on
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    import image_captioning_amd as amd
    from image_captioning_amd import distributed as D

    # RCCL ("nccl") is the backend of every real run; MIC_DIST_BACKEND=gloo lets the multi-rank control flow be
    # rehearsed on a box with fewer GPUs than ranks (ranks then share devices round-robin)
    backend = os.environ.get("MIC_DIST_BACKEND", "nccl")
    rank, local, world = D.init_from_env(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    L = amd._lib

    wl = WORKLOADS[args.workload]
    beam = wl["beam"]
    sd = build_state(args.workload)
    vocab = amd.Vocabulary.synthetic(V)
    if args.workload == "lstm":
        model = amd.lstm.IntermediateImageCaptioner(vocab, torch.bfloat16, sync_every=MAX_LEN).load_state_dict(sd).to(dev)
    elif args.workload == "basic":
        model = amd.lstm.BasicImageCaptioner(vocab, torch.float32, sync_every=MAX_LEN).load_state_dict(sd).to(dev)
    else:
        model = amd.transformer.ImageCaptioner(vocab, HEADS, torch.bfloat16, encoder="swin" if args.workload in ("swin", "swin_t") else "resnet50",
                                               sync_every=MAX_LEN, decode_groups=args.decode_groups).load_state_dict(sd).to(dev)
    if args.workload == "job":
        return run_job(args, amd, D, model, sd, dev, rank, world, wl)
    B = args.batch or wl["batch"]
    images = torch.randn(B, 3, IMG, IMG, generator=torch.Generator(device=dev).manual_seed(1234 + rank), device=dev)
    n_total = B * world

    n_lanes = max(1, args.lanes)
    # one pool for the timed region and the 3-lane measurement: more streams than hardware queues would make lanes take turns
    lanes = amd.lanes.Lanes(model, max(3, n_lanes) if args.concurrent_steps > 0 and world == 1 else n_lanes)

    def caption(m, _item):
        if args.workload == "basic":  # batched greedy decoding (models/image_captioner.py:82-105); no scores, PAD after EOS
            t = m.sample_ids(m.encoder(images), MAX_LEN, None)
            tokens = torch.nn.functional.pad(t, (0, MAX_LEN + 1 - t.shape[1]), value=-1)
            return tokens, torch.full((B,), t.shape[1], dtype=torch.int32, device=dev), torch.zeros(B, device=dev)
        if args.workload == "lstm":
            tokens, lengths = m.beam_search_ids(m.encoder(images), MAX_LEN, beam)
            return tokens, lengths, torch.zeros(B, device=dev)  # models/image_captioner.py:61-69 returns no scores
        tokens, lengths, score, _ = m.beam_search_ids(m.encode(images), MAX_LEN, beam)
        return tokens, lengths, score

    def run_steps(n, in_flight=n_lanes):
        """n steps = n whole batches through encoder + search, at most `in_flight` of them at a time; then ONE gather
        across ranks of the ids of all n steps (nothing is exchanged while the steps run)."""
        if n == 0:
            return None
        parts = [tuple(x.clone() for x in r) for r in lanes.map(caption, range(n), in_flight)]
        tokens, lengths, score = (torch.cat([q[i] for q in parts]) for i in range(3))
        return D.gather_captions(tokens, lengths, score, n * n_total)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(len(lanes), len(lanes))  # set-up, not a step: every lane allocates its workspaces and captures its graphs
    run_steps(args.warmup)
    fence()
    L.check(L.lib.mic_profile_begin(max(1, args.steps) * 1200), "mic_profile_begin")
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    cnt, ms, fl = (C.c_int32 * 4)(), (C.c_double * 4)(), (C.c_double * 4)()
    L.check(L.lib.mic_profile_end(cnt, ms, fl), "mic_profile_end")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    executed_steps = int(out[1].max().item()) - 1  # generated tokens of the longest caption

    concurrent = None
    if world == 1 and n_lanes == 1 and args.concurrent_steps > 0:
        # throughput with three whole batches in flight, outside the timed region: the timed region stays one batch at a
        # time so that its per-kernel durations are those of a kernel that has the GPU to itself
        n3 = args.concurrent_steps
        run_steps(3, 3)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run_steps(n3, 3)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - ts
        concurrent = {"lanes": 3, "steps": n3, "value": round(B * n3 / dt3, 2), "unit": "images/s", "ms_per_step": round(dt3 / n3 * 1e3, 3)}

    single = None
    if world == 1 and n_lanes > 1:
        # the same kernels with the GPU to themselves: under concurrent lanes a launch's duration includes the other
        # lanes' work, so the per-kernel reading is taken again on one lane, outside the timed region
        run_steps(1, 1)
        torch.cuda.synchronize()
        L.check(L.lib.mic_profile_begin(3 * 1200), "mic_profile_begin")
        ts = time.perf_counter()
        run_steps(3, 1)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts) / 3 * 1e3
        c1, m1, f1 = (C.c_int32 * 4)(), (C.c_double * 4)(), (C.c_double * 4)()
        L.check(L.lib.mic_profile_end(c1, m1, f1), "mic_profile_end")
        single = (c1[1], m1[1], f1[1], single_ms)

    decode, schedule = None, None
    if rank == 0 and args.workload != "basic":
        # the search alone (cross-K/V or initial state + beam steps + best-beam extraction) between two events on the
        # launch stream, outside the timed region and with the GPU to itself like the timed region's batches, priced
        # against HBM bandwidth by its algorithmic bytes
        lstm = args.workload == "lstm"
        feats = model.encoder(images) if lstm else model.encode(images)
        model.beam_search_ids(feats, MAX_LEN, beam)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            model.beam_search_ids(feats, MAX_LEN, beam)
        e1.record()
        torch.cuda.synchronize()
        d_ms = e0.elapsed_time(e1) / reps
        decode = lstm_decode_roofline(d_ms, B, beam, executed_steps) if lstm else decode_roofline(d_ms, B, beam, executed_steps, feats.shape[1])
        if not lstm:
            schedule = model.last_schedule
            decode["step_schedule"] = schedule
    if hasattr(model, "check_errors"):
        model.check_errors()  # raises if a workgroup of a one-launch decode step gave up waiting for another

    if rank == 0:
        # dominant kernel: igemm_kernel<bf16> with tiles of 128 rows and more (slot 1), the encoder's implicit-GEMM convolutions
        # (together with the fused bottleneck launches, which share the slot); the fp32 workload's are in slot 0
        ks = 0 if args.workload == "basic" else 1
        k_ms, k_n, k_fl = ms[ks], cnt[ks], fl[ks]
        achieved = (k_fl / k_n) / (k_ms / k_n * 1e-3) / 1e12 if k_n else 0.0
        peak = MFMA_F32_PEAK_TFLOPS if ks == 0 else MFMA_BF16_PEAK_TFLOPS
        traffic, traffic_file = measured_traffic()

        def igemm_roofline(n, t_ms, flops):
            tf = flops / (t_ms * 1e-3) / 1e12 if n else 0.0
            return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "avg_launch_us": round(t_ms / n * 1e3, 2) if n else None}
        res = {
            "metric": f"images captioned/sec (224^2, beam={beam}, max_len=30)", "value": round(n_total * args.steps / elapsed, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.workload == "basic" else "bf16", "data": "synthetic",
            "config": {"workload": wl["label"] if B == wl["batch"] else wl["label"].replace(f"batch {wl['batch']}", f"batch {B}"),
                       "images_per_gpu_per_step": B, "executed_decode_steps": executed_steps, "parallelism": f"dp{world}",
                       "lanes_per_gpu": n_lanes, "decode_step_schedule": schedule},
            "roofline": {"bound": "mfma", "kernel": ("igemm_kernel<f32>" if ks == 0 else "igemm_kernel<bf16>, fused_block_kernel and stem_pool_kernel") +
                         " (every launch of the encoder trunk and of the cross-K/V projection: one measurement scope around "
                         "each, whatever the tile shape)", "achieved": round(achieved, 2),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic if args.workload == "transformer" and B == BATCH else None,
                         "traffic_unit": f"bytes/launch (rocprofv3 PMC, profiles/{traffic_file})",
                         "launches_per_step": k_n // max(1, args.steps),
                         "avg_launch_us": round(k_ms / k_n * 1e3, 2) if k_n else None,
                         "gflop_per_launch": round(k_fl / k_n / 1e9, 3) if k_n else None,
                         },
        }
        if cnt[3]:  # GEMM launches outside the measurement scopes (inside a scope every launch counts towards the scope)
            res["roofline"]["other_gemm_launches"] = {"igemm<bf16,64-row tiles>": {"n": cnt[3], "ms": round(ms[3], 3),
                                                      "tflops": round(fl[3] / (ms[3] * 1e-3) / 1e12, 2) if ms[3] else None}}
        if single is not None:
            res["roofline"]["note"] = (f"launch durations in the timed region, {n_lanes} batches in flight: a launch shares the GPU with "
                                       "the other lanes' kernels; roofline_single_lane is the same kernel with the GPU to itself")
            res["roofline_single_lane"] = dict(igemm_roofline(*single[:3]), ms_per_step=round(single[3], 3),
                                               images_per_s=round(B / (single[3] * 1e-3), 1))
        if concurrent is not None:
            res["concurrent_lanes"] = concurrent
        if decode is not None:
            res["roofline_decode"] = decode
        if args.cpu_images > 0 and world == 1:
            # configs[0] is timed in full (its batch of 8); the others on a sample of the same workload
            res["cpu_baseline"] = cpu_baseline(sd, min(args.cpu_images, B) if args.workload == "basic" else args.cpu_images, args.workload, beam)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
