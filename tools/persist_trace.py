"""Phase timeline of the one-launch decoder step: where the workgroups of a step spend their time (100 MHz stamps)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
import image_captioning_amd as amd  # noqa: E402

L = amd._lib


def main():
    dev = torch.device("cuda:0")
    V, H, Ly, S, B, K, T = 3500, 512, 3, 49, 256, 3, 30
    sd = amd.weights.random_transformer_state(V, H, Ly, 40, seed=0)
    vocab = amd.Vocabulary.synthetic(V)
    feats = torch.randn(B, S, H, generator=torch.Generator().manual_seed(1)).relu().to(dev, torch.bfloat16)
    m = amd.transformer.ImageCaptioner(vocab, 2, torch.bfloat16, encoder="none", sync_every=T, use_graph=False).load_state_dict(sd).to(dev)
    for _ in range(2):
        m.beam_search_ids(feats, T, K)
    torch.cuda.synchronize()
    trace = torch.zeros(512 * 128, dtype=torch.int64, device=dev)
    L.check(L.lib.mic_tr_persist_trace(trace.data_ptr()), "trace")
    m.beam_search_ids(feats, T, K)  # the buffer keeps the stamps of the last step (t = 29: the longest prefix)
    torch.cuda.synchronize()
    L.check(L.lib.mic_tr_persist_trace(None), "trace")
    m.check_errors()
    tr = trace.cpu().view(512, 128)[:256].double() / 100.0  # microseconds
    t0 = tr[:, 0].min()
    names = []
    for l in range(Ly):
        names += [f"L{l} in-proj", f"L{l} attention", f"L{l} ff1", f"L{l} ff2"]
    names.append("vocabulary")
    print(f"workgroup start spread: {(tr[:, 0].max() - t0):.2f} us")
    print(f"{'phase':14s} {'active':>6s} {'ready(mean/max)':>18s} {'work mean':>10s} {'work max':>9s} {'done max':>9s} {'arrive max':>10s}")
    prev_done = 0.0
    for p, name in enumerate(names):
        ready, done, arr = tr[:, 3 * p + 1], tr[:, 3 * p + 2], tr[:, 3 * p + 3]
        act = ready > 0
        r = ready[act] - t0
        d = done[act] - t0
        work = (done - ready)[act]
        allarr = arr[arr > 0] - t0 if p < len(names) - 1 else torch.tensor([0.0])
        print(f"{name:14s} {int(act.sum()):6d} {r.mean():9.2f}/{r.max():8.2f} {work.mean():10.2f} {work.max():9.2f} {d.max():9.2f} {allarr.max():10.2f}")
    if (tr[:, 121] > 0).any():
        act = tr[:, 120] > 0
        print(f"{'beam tail':14s} {int(act.sum()):6d} {(tr[act, 120] - t0).mean():9.2f}/{(tr[act, 120] - t0).max():8.2f} "
              f"{(tr[act, 121] - tr[act, 120]).mean():10.2f} {(tr[act, 121] - tr[act, 120]).max():9.2f} {(tr[act, 121] - t0).max():9.2f}")
    print(f"kernel span: {(tr[:, 1:].max() - t0):.2f} us")
    print("inside the GEMM phases, microseconds after 'ready' (mean over active workgroups): stats merged, rows loaded, rows staged, products done, epilogue values, statistics, outputs stored")
    for p, name in enumerate(names):
        if "attention" in name:
            continue
        ready, done = tr[:, 3 * p + 1], tr[:, 3 * p + 2]
        act = ready > 0
        parts = []
        for k in range(6):
            v = tr[:, 40 + 6 * p + k]
            ok = act & (v > 0)
            parts.append(f"{(v - ready)[ok].mean():6.2f}" if ok.any() else "     -")
        parts.append(f"{(done - ready)[act].mean():6.2f}")
        print(f"{name:14s} " + " ".join(parts))


if __name__ == "__main__":
    main()
