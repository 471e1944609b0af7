"""Search phase alone (cross-K/V + 30 beam steps + best-beam extraction) at the benchmark's decode shape, timed with
events around whole searches: the one-launch step against one launch per sub-layer, interleaved in one process."""
import argparse
import sys

import torch

sys.path.insert(0, ".")
import image_captioning_amd as amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--max-len", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    V, H, Ly, S = 3500, 512, 3, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, 40, seed=0)
    vocab = amd.Vocabulary.synthetic(V)
    feats = torch.randn(args.batch, S, H, generator=torch.Generator().manual_seed(1)).relu().to(dev, torch.bfloat16)
    models = {}
    for name, persistent in (("launch-per-sublayer", False), ("one-launch", True)):
        models[name] = amd.transformer.ImageCaptioner(vocab, 2, torch.bfloat16, encoder="none", sync_every=args.max_len,
                                                      persistent=persistent).load_state_dict(sd).to(dev)
    out = {}
    for name, m in models.items():
        for _ in range(3):
            out[name] = m.beam_search_ids(feats, args.max_len, args.beam)
        torch.cuda.synchronize()
        m.check_errors()
        print(name, "schedule:", m.last_schedule)
    same = all(torch.equal(a, b) for a, b in zip(out["launch-per-sublayer"][:3], out["one-launch"][:3]))
    print("ids, lengths and scores identical:", same)
    for r in range(args.rounds):
        for name, m in models.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                m.beam_search_ids(feats, args.max_len, args.beam)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print(f"round {r} {name:22s} {ms:7.3f} ms per search  ({ms / args.max_len * 1e3:6.1f} us per step)")
    for m in models.values():
        m.check_errors()


if __name__ == "__main__":
    main()
