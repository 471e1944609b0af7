"""GPU parity tests for the gaps a review of the first test suite named: fp32 search at the benchmark's
model size, every tile variant of the convolution kernel by name, the padded-weight contract, the
graph cache under eviction with two lanes in flight, a NaN score in beam selection, and a checkpoint
file taken all the way to captions on the device."""
import ctypes as C
import math
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32_TOL = dict(rtol=1e-3, atol=1e-3)


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


class IdVocab:
    """ids 0-3 special, every other id is its own decimal string (the search is checked on ids)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def str_to_idx(self, s):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[s]

    def encode_as_words(self, ids):
        return [int(i) for i in ids]


# ------------------------------------------------------------------------------------------------
# fp32 search at the benchmark's model size (models/transformer.py:458-640)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("beam", [1, 3])
def test_transformer_fp32_benchmark_model_size(amd, beam):
    """H=512, 2 heads, 3 layers, V=3500, S=49, max_len=30 in fp32 against the fp32 CPU oracle: token ids
    identical, summed log-probabilities within 1e-3.  These are the shapes that select the 2-KiB-row attention
    kernels, the K=512/1024 fp32 weight-streaming GEMMs and the in-register top-k at V=3500."""
    from oracle import captioner_oracle as O
    V, H, Ly, PT, B, S, heads, T = 3500, 512, 3, 40, 6, 49, 2, 30
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=21)
    g = torch.Generator().manual_seed(9)
    for k in list(sd):  # biases and LayerNorm parameters away from their trivial initial values
        if k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    vocab = IdVocab(V)
    model = amd.transformer.ImageCaptioner.from_state_dict(sd, vocab, heads, torch.float32, DEV)
    feats_bchw = torch.randn(B, H, 7, 7, generator=g).relu()
    feats = O.tokens_major(feats_bchw)
    ref_tok, ref_lp, _ = O.transformer_generate(sd, heads, feats_bchw, T, None, beam)
    caps, lp = model.generate_from_features(feats, vocab, T, None, beam)
    assert caps == [vocab.encode_as_words(t) for t in ref_tok], f"beam {beam}"
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), **F32_TOL)


# ------------------------------------------------------------------------------------------------
# every tile variant of the convolution kernel, by name
# ------------------------------------------------------------------------------------------------
def _conv_variant(L, W, variant, x_nchw, w, bias, stride, pad, relu, dtype, res_nchw=None):
    B, Ci, H, Wd = x_nchw.shape
    Co, _, kh, kw = w.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wm = W.conv_matrix(w, dtype, DEV)
    out = torch.empty(B, OH, OW, Co, dtype=dtype, device=DEV)
    res = None if res_nchw is None else res_nchw.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    b = bias.to(DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.res, d.out = x.data_ptr(), wm.data_ptr(), b.data_ptr(), L.ptr(res), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.dtype_code(dtype)
    d.M, d.N, d.K = B * OH * OW, Co, kh * kw * Ci
    d.OH, d.OW, d.IH, d.IW = OH, OW, H, Wd
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = H * Wd * Ci, Wd * Ci, Ci
    d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = kw, Ci, stride, pad, pad
    d.ldc, d.ld_res, d.relu, d.w_rows = Co, Co, int(relu), wm.shape[0]
    L.check(L.lib.mic_conv2d_variant(C.byref(d), variant, L.stream_ptr()), f"mic_conv2d_variant({variant})")
    return out.float().cpu().permute(0, 3, 1, 2)


VARIANT_CASES = [
    # B, Ci, H, W, Co, k, stride, pad, residual:   M and N tails against 64-, 128- and 256-wide tiles
    (3, 64, 13, 11, 320, 1, 1, 0, True),     # M = 429, N = 320: tails in both for every tile size
    (2, 64, 14, 14, 96, 3, 1, 1, False),     # 3x3, K = 576 (18 k-tiles of 32: odd count for two-tile stages), N < one tile
    (2, 128, 15, 15, 256, 3, 2, 1, False),   # stride 2, K = 1152, N = one 256-wide tile exactly
    (1, 96, 9, 9, 130, 1, 1, 0, True),       # K = 96 (3 k-tiles), N two rows past a 128 boundary, M = 81 < any tile
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17])
def test_conv2d_every_tile_variant(amd, variant, dtype):
    L, W = amd._lib, amd.weights
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for case in VARIANT_CASES:
        B, Ci, H, Wd, Co, k, s, p, use_res = case
        g = torch.Generator().manual_seed(sum(case) + variant)
        x = torch.randn(B, Ci, H, Wd, generator=g)
        w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
        b = torch.randn(Co, generator=g)
        if dtype == torch.bfloat16:
            x, w = x.bfloat16().float(), w.bfloat16().float()
        ref = F.conv2d(x, w, b, s, p)
        res = None
        if use_res:
            res = torch.randn(ref.shape, generator=g)
            if dtype == torch.bfloat16:
                res = res.bfloat16().float()
            ref = ref + res
        ref = F.relu(ref)
        got = _conv_variant(L, W, variant, x, w, b, s, p, True, dtype, res)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), err_msg=f"variant {variant} case {case}", **tol)


EIGHT_PHASE_CASES = [
    # B, Ci, H, W, Co, k, stride, pad, residual: N a multiple of 256 and channels a multiple of 64, as variant 18 requires
    (3, 64, 13, 11, 256, 1, 1, 0, True),     # one K-tile, M = 429: a partial second row tile
    (2, 64, 14, 14, 512, 3, 1, 1, False),    # 9 K-tiles (odd count), taps outside the image
    (2, 128, 15, 15, 256, 3, 2, 1, False),   # stride 2, 18 K-tiles
    (1, 192, 9, 9, 256, 1, 1, 0, True),      # 3 K-tiles, M = 81
    (5, 256, 14, 14, 768, 3, 1, 1, True),    # several tiles in both directions, 36 K-tiles
    (3, 512, 7, 7, 256, 3, 1, 1, False),     # one tile, 72 K-tiles: split three ways when a scratch is given
]


@pytest.mark.parametrize("splitk", ["none", "auto", "4"])
def test_conv2d_eight_phase_variant(amd, splitk, monkeypatch):
    """Tile variant 18 (256x256, eight phases, bf16): against torch's fp32 convolution, bit for bit against variant 0 when K
    is not split, within one bf16 rounding step of it when it is (the fp32 sum is regrouped); the tickets of the scratch are
    zero again after every launch, and two launches give the same bits."""
    L, W = amd._lib, amd.weights
    if splitk == "4":
        monkeypatch.setenv("MIC_IGEMM8_SPLITK", "4")
    scratch = torch.zeros(int(L.lib.mic_conv2d_scratch_bytes()), dtype=torch.uint8, device=DEV) if splitk != "none" else None
    for case in EIGHT_PHASE_CASES:
        B, Ci, H, Wd, Co, k, s, p, use_res = case
        g = torch.Generator().manual_seed(sum(case))
        x = torch.randn(B, Ci, H, Wd, generator=g).bfloat16().float()
        w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).bfloat16().float()
        b = torch.randn(Co, generator=g)
        ref = F.conv2d(x, w, b, s, p)
        res = None
        if use_res:
            res = torch.randn(ref.shape, generator=g).bfloat16().float()
            ref = ref + res
        ref = F.relu(ref)
        outs = []
        for variant, scr in ((0, None), (18, scratch), (18, scratch)):
            OH, OW = (H + 2 * p - k) // s + 1, (Wd + 2 * p - k) // s + 1
            xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
            wm = W.conv_matrix(w, torch.bfloat16, DEV)
            out = torch.empty(B, OH, OW, Co, dtype=torch.bfloat16, device=DEV)
            rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
            bd = b.to(DEV)
            d = L.ConvDesc()
            d.in_, d.w, d.bias, d.res, d.out = xd.data_ptr(), wm.data_ptr(), bd.data_ptr(), L.ptr(rd), out.data_ptr()
            d.dtype = d.out_dtype = d.res_dtype = L.BF16
            d.M, d.N, d.K = B * OH * OW, Co, k * k * Ci
            d.OH, d.OW, d.IH, d.IW = OH, OW, H, Wd
            d.in_img_stride, d.in_row_stride, d.in_pix_stride = H * Wd * Ci, Wd * Ci, Ci
            d.kw_taps, d.run, d.stride, d.pad_h, d.pad_w = k, Ci, s, p, p
            d.ldc, d.ld_res, d.relu, d.w_rows = Co, Co, 1, wm.shape[0]
            if scr is not None:
                d.scratch, d.scratch_bytes = scr.data_ptr(), scr.numel()
            L.check(L.lib.mic_conv2d_variant(C.byref(d), variant, L.stream_ptr()), f"mic_conv2d_variant({variant})")
            torch.cuda.synchronize()
            outs.append(out)
        v0, a18, b18 = outs
        np.testing.assert_allclose(a18.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-2, atol=2e-2, err_msg=f"case {case}")
        assert torch.equal(a18, b18), f"two launches differ, case {case}"
        if splitk == "none":
            assert torch.equal(a18, v0), f"variant 18 differs from variant 0, case {case}"
        else:
            np.testing.assert_allclose(a18.float().cpu().numpy(), v0.float().cpu().numpy(), rtol=8e-3, atol=1e-3, err_msg=f"case {case}")
            assert int(scratch[:4096].to(torch.int32).sum().item()) == 0, "tickets left behind"
    # shapes the variant does not take are refused, not mis-computed
    x = torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16)
    wm = torch.zeros(128, 64, device=DEV, dtype=torch.bfloat16)
    out = torch.zeros(64, 128, device=DEV, dtype=torch.bfloat16)
    d = L.ConvDesc()
    d.in_, d.w, d.out = x.data_ptr(), wm.data_ptr(), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = 64, 128, 64, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc, d.w_rows = 64, 64, 1, 64, 1, 128, 128
    assert L.lib.mic_conv2d_variant(C.byref(d), 18, L.stream_ptr()) == -1 and b"eight-phase" in L.lib.mic_last_error()
    torch.cuda.synchronize()


def test_conv2d_rejects_a_weight_matrix_that_is_not_padded(amd):
    """Tiles fetch whole 128-row groups of the weight matrix: a matrix stated to be shorter is refused."""
    L = amd._lib
    x = torch.zeros(64, 64, device=DEV)
    w = torch.zeros(96, 64, device=DEV)   # N = 96 rows, not padded to 128
    out = torch.zeros(64, 96, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.F32
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = 64, 96, 64, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc = 64, 4, 1, 64, 1, 96
    d.w_rows = 96
    for v in (-1, 0, 16):
        rc = L.lib.mic_conv2d(C.byref(d), L.stream_ptr()) if v < 0 else L.lib.mic_conv2d_variant(C.byref(d), v, L.stream_ptr())
        assert rc == -1 and b"zero-padded to 128" in L.lib.mic_last_error()
    assert L.lib.mic_conv2d_variant(C.byref(d), 19, L.stream_ptr()) == -1
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# graph cache: more distinct step runs than the cache holds, two lanes launching at once
# ------------------------------------------------------------------------------------------------
def test_graph_cache_eviction_with_two_lanes(amd):
    """Each (t0, t1) run of beam steps is its own captured graph.  Two host threads on two streams walk 30 steps
    in runs of one step each, twice (60 distinct keys per lane counting both lanes' workspaces, against a cache of
    32), so executables are dropped while the other lane has launches in flight; the captions must equal those of
    the uncaptured path."""
    L = amd._lib
    V, H, Ly, PT, B, S, heads, T = 300, 64, 2, 40, 4, 49, 2, 30
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=5)
    vocab = IdVocab(V)
    plain = amd.transformer.ImageCaptioner(vocab, heads, torch.float32, encoder="none", sync_every=T, use_graph=False)
    plain.load_state_dict(sd).to(DEV)
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(B, S, H, generator=g).relu() for _ in range(2)]
    want = [[t.clone() for t in plain.beam_search_ids(f, T, 3)[:3]] for f in feats]
    graphed = amd.transformer.ImageCaptioner(vocab, heads, torch.float32, encoder="none", sync_every=1, use_graph=True)
    graphed.load_state_dict(sd).to(DEV)
    lanes = amd.lanes.Lanes(graphed, 2)

    def run(m, i):
        out = None
        for _ in range(2):
            out = [t.clone() for t in m.beam_search_ids(feats[i], T, 3)[:3]]
        return out

    got = lanes.map(run, [0, 1])
    torch.cuda.synchronize()
    for w, r in zip(want, got):
        for a, b in zip(w, r):
            assert torch.equal(a, b)
    cached, parked = C.c_int32(), C.c_int32()
    L.check(L.lib.mic_graph_cache_stats(C.byref(cached), C.byref(parked)))
    assert 1 <= cached.value <= 32
    assert parked.value == 0  # everything has finished, so every dropped executable has been freed


# ------------------------------------------------------------------------------------------------
# beam selection with a NaN candidate stays in bounds
# ------------------------------------------------------------------------------------------------
def test_beam_select_with_nan_scores_stays_in_bounds(amd):
    L = amd._lib
    B, K, t_cap, t = 3, 3, 8, 2
    N = B * K

    def state():
        score = torch.zeros(N, dtype=torch.float64, device=DEV)
        ln = torch.full((N,), t + 1, dtype=torch.int32, device=DEV)
        seq = torch.full((N, t_cap), 5, dtype=torch.int32, device=DEV)
        anc = torch.zeros(N, t_cap, dtype=torch.int32, device=DEV)
        st = L.BeamState()
        st.score, st.len, st.seq, st.anc = score.data_ptr(), ln.data_ptr(), seq.data_ptr(), anc.data_ptr()
        return st, (score, ln, seq, anc)

    cur, cb = state()
    nxt, nb = state()
    nb[2].fill_(-7)
    nb[1].fill_(-7)
    top_val = torch.log(torch.rand(N, K, generator=torch.Generator().manual_seed(1))).to(DEV)
    top_val[0:K] = float("nan")      # image 0: every candidate NaN
    top_val[K, 1] = float("nan")     # image 1: one NaN among finite candidates
    top_idx = torch.randint(4, 50, (N, K), generator=torch.Generator().manual_seed(2), dtype=torch.int32).to(DEV)
    parent = torch.full((N,), -1, dtype=torch.int32, device=DEV)
    live = torch.zeros(1, dtype=torch.int32, device=DEV)
    sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64, device=DEV)
    L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), top_val.data_ptr(), top_idx.data_ptr(), parent.data_ptr(), live.data_ptr(),
                                  sqrt_tab.data_ptr(), B, K, K, t, t_cap, 2, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    par, ln, seq = parent.cpu(), nb[1].cpu(), nb[2].cpu()
    for b in range(B):
        assert all(b * K <= int(p) < (b + 1) * K for p in par[b * K:(b + 1) * K]), "parent outside its image"
    assert bool((ln == t + 2).all()) and int(seq[:, : t + 2].min()) >= 4
    # image 1: the NaN candidate sorts last, so no survivor carries its token unless nothing else is left
    fin = top_val[K:2 * K].cpu()
    keys = [(float(fin[i, j]), i, j) for i in range(K) for j in range(K) if not math.isnan(float(fin[i, j]))]
    keys.sort(key=lambda e: (-e[0], e[1] * K + e[2]))
    want = [int(top_idx[K + i, j]) for _, i, j in keys[:K]]
    assert [int(seq[K + r, t + 1]) for r in range(K)] == want
    # image 2 (no NaN anywhere) is the ordinary stable descending selection
    fin = top_val[2 * K:3 * K].cpu()
    keys = sorted(((float(fin[i, j]), i, j) for i in range(K) for j in range(K)), key=lambda e: (-e[0], e[1] * K + e[2]))
    assert [int(seq[2 * K + r, t + 1]) for r in range(K)] == [int(top_idx[2 * K + i, j]) for _, i, j in keys[:K]]


# ------------------------------------------------------------------------------------------------
# a checkpoint FILE in the reference's layout -> captions on the device
# ------------------------------------------------------------------------------------------------
def test_checkpoint_file_to_captions_on_the_device(amd, tmp_path):
    """train.save_checkpoint's dictionary (train.py:390-403) written with torch.save, read back by load_checkpoint
    (the counterpart of plotter/captions.py:386-397), then gen_caption on images: ids equal the oracle's on the
    same state dict, transformer and intermediate models."""
    from oracle import captioner_oracle as O
    V, H, Ly, heads, PT = 300, 64, 2, 2, 20
    vocab = amd.Vocabulary.synthetic(V)
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(8))
    sd = amd.weights.random_resnet50_state(0)
    sd.update(amd.weights.random_head_state("transformer", H, 1))
    sd.update(amd.weights.random_transformer_state(V, H, Ly, PT, seed=3))
    config = {"model": "transformer", "encoder": "resnet50", "num_heads": heads, "hidden_size": H, "num_layers": Ly,
              "max_caption_len": PT, "beam_size": 3}
    path = tmp_path / "BEST_transformer.pt"
    torch.save({"epoch": 3, "model_state": sd, "optimizer_state": {}, "scheduler_state": None, "best_val_loss": 2.4,
                "train_loss": 2.1, "epochs_no_improve": 0, "lr": 1e-4, "config": config, "use_rl": False}, path)
    model, cfg = amd.checkpoint.load_checkpoint(str(path), vocab, torch.float32)
    assert cfg["num_heads"] == heads
    with torch.no_grad():
        feats = O.encoder_transformer(sd, img)
    for K in (3, 1):
        ref_tok, ref_lp, _ = O.transformer_generate(sd, heads, feats, 10, None, K)
        caps, lp = amd.gen_caption(model, img, vocab, 10, torch.device(DEV), None, K)
        assert caps == [vocab.encode_as_words(t) for t in ref_tok]
        np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), **F32_TOL)

    sd2 = amd.weights.random_resnet50_state(1)
    sd2.update(amd.weights.random_head_state("intermediate", H, 2))
    sd2.update(amd.weights.random_lstm_state("intermediate", V, H, H, 2, seed=4))
    path2 = tmp_path / "LAST_intermediate.pt"
    torch.save({"epoch": 1, "model_state": sd2, "config": {"model": "intermediate", "encoder": "resnet50"}, "use_rl": False}, path2)
    model2, _ = amd.checkpoint.load_checkpoint(str(path2), vocab, torch.float32)
    with torch.no_grad():
        f2 = O.encoder_intermediate(sd2, img)
    for K in (3, 1):
        ref = O.lstm_generate(sd2, "intermediate", f2, 10, None, K)
        caps, none = amd.gen_caption(model2, img, vocab, 10, torch.device(DEV), None, K)
        assert none is None and caps == [vocab.encode_as_words(t) for t in ref]


# ------------------------------------------------------------------------------------------------
# fused bottleneck tail + next head (torchvision Bottleneck.forward via models/encoders/transformer.py:46)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Cm,C2,B,H,Wd", [(64, 0, 3, 14, 14), (64, 64, 3, 14, 14), (64, 128, 2, 9, 11), (128, 0, 2, 9, 11),
                                         (128, 128, 3, 14, 14), (64, 64, 1, 5, 5), (128, 128, 5, 28, 28)])
def test_bottleneck_fused_matches_torch(amd, Cm, C2, B, H, Wd):
    """conv3x3+ReLU -> conv1x1 + residual + ReLU -> next conv1x1 + ReLU in one launch against the same chain in torch
    fp32 with bf16 rounding at the same three points (what the layer-by-layer path stores)."""
    L, W = amd._lib, amd.weights
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(Cm + C2 + H)
    r16 = lambda t: t.to(bf).float()
    t1 = r16(torch.randn(B, Cm, H, Wd, generator=g).relu())
    x = r16(torch.randn(B, 4 * Cm, H, Wd, generator=g).relu())
    w2 = r16(torch.randn(Cm, Cm, 3, 3, generator=g) / math.sqrt(9 * Cm))
    w3 = r16(torch.randn(4 * Cm, Cm, 1, 1, generator=g) / math.sqrt(Cm))
    b2, b3 = 0.1 * torch.randn(Cm, generator=g), 0.1 * torch.randn(4 * Cm, generator=g)
    t2 = r16(F.relu(F.conv2d(t1, w2, b2, 1, 1)))
    xo = r16(F.relu(F.conv2d(t2, w3, b3) + x))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, bf)
    d = L.FusedBlockDesc()
    keep = [nhwc(t1), W.conv_matrix(w2, bf, DEV), b2.to(DEV), W.conv_matrix(w3, bf, DEV), b3.to(DEV), nhwc(x),
            torch.full((B, H, Wd, 4 * Cm), 7.0, dtype=bf, device=DEV)]
    d.t1, d.w2, d.b2, d.w3, d.b3, d.res, d.xout = (t.data_ptr() for t in keep)
    d.dtype, d.B, d.H, d.W, d.C, d.C2 = L.BF16, B, H, Wd, Cm, C2
    d.w2_rows, d.w3_rows = keep[1].shape[0], keep[3].shape[0]
    if C2:
        w1n = r16(torch.randn(C2, 4 * Cm, 1, 1, generator=g) / math.sqrt(4 * Cm))
        b1n = 0.1 * torch.randn(C2, generator=g)
        tn = r16(F.relu(F.conv2d(xo, w1n, b1n)))
        keep += [W.conv_matrix(w1n, bf, DEV), b1n.to(DEV), torch.full((B, H, Wd, C2), 7.0, dtype=bf, device=DEV)]
        d.w1n, d.b1n, d.t1n = (t.data_ptr() for t in keep[7:])
        d.w1n_rows = keep[7].shape[0]
    L.check(L.lib.mic_bottleneck_fused(C.byref(d), L.stream_ptr()), "mic_bottleneck_fused")
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(keep[6].float().cpu().permute(0, 3, 1, 2).numpy(), xo.numpy(), **tol)
    if C2:
        np.testing.assert_allclose(keep[9].float().cpu().permute(0, 3, 1, 2).numpy(), tn.numpy(), **tol)


@pytest.mark.parametrize("B,H,Wd", [(3, 14, 14), (1, 5, 7), (4, 56, 56)])
def test_bottleneck_fused_with_shortcut_convolution(amd, B, H, Wd):
    """The first block of ResNet-50's first stage in one launch: conv3x3+ReLU -> [conv1x1 | shortcut conv1x1 of the block's
    input] + ReLU -> next conv1x1 + ReLU, against the same chain in torch with bf16 rounding at the stored points."""
    L, W = amd._lib, amd.weights
    bf = torch.bfloat16
    Cm = C2 = CS = 64
    assert L.lib.mic_bottleneck_fused_shortcut_supported(L.BF16, 64, 64, 64) == 1
    assert L.lib.mic_bottleneck_fused_shortcut_supported(L.BF16, 128, 128, 256) == 0
    g = torch.Generator().manual_seed(B + H)
    r16 = lambda t: t.to(bf).float()
    t1 = r16(torch.randn(B, Cm, H, Wd, generator=g).relu())
    x = r16(torch.randn(B, CS, H, Wd, generator=g).relu())
    w2 = r16(torch.randn(Cm, Cm, 3, 3, generator=g) / math.sqrt(9 * Cm))
    w3 = r16(torch.randn(4 * Cm, Cm, 1, 1, generator=g) / math.sqrt(Cm))
    wd = r16(torch.randn(4 * Cm, CS, 1, 1, generator=g) / math.sqrt(CS))
    b2, b3 = 0.1 * torch.randn(Cm, generator=g), 0.1 * torch.randn(4 * Cm, generator=g)
    w1n = r16(torch.randn(C2, 4 * Cm, 1, 1, generator=g) / math.sqrt(4 * Cm))
    b1n = 0.1 * torch.randn(C2, generator=g)
    t2 = r16(F.relu(F.conv2d(t1, w2, b2, 1, 1)))
    xo = r16(F.relu(F.conv2d(t2, w3, b3) + F.conv2d(x, wd)))
    tn = r16(F.relu(F.conv2d(xo, w1n, b1n)))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, bf)
    w3d = W.conv_matrix(torch.cat([w3, wd], dim=1), bf, DEV)   # [4C][C + CS]
    keep = [nhwc(t1), W.conv_matrix(w2, bf, DEV), b2.to(DEV), w3d, b3.to(DEV), nhwc(x),
            torch.full((B, H, Wd, 4 * Cm), 7.0, dtype=bf, device=DEV), W.conv_matrix(w1n, bf, DEV), b1n.to(DEV),
            torch.full((B, H, Wd, C2), 7.0, dtype=bf, device=DEV)]
    d = L.FusedBlockDesc()
    d.t1, d.w2, d.b2, d.w3, d.b3, d.xs, d.xout, d.w1n, d.b1n, d.t1n = (t.data_ptr() for t in keep)
    d.dtype, d.B, d.H, d.W, d.C, d.C2, d.CS = L.BF16, B, H, Wd, Cm, C2, CS
    d.w2_rows, d.w3_rows, d.w1n_rows = keep[1].shape[0], keep[3].shape[0], keep[7].shape[0]
    L.check(L.lib.mic_bottleneck_fused(C.byref(d), L.stream_ptr()), "mic_bottleneck_fused")
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(keep[6].float().cpu().permute(0, 3, 1, 2).numpy(), xo.numpy(), **tol)
    np.testing.assert_allclose(keep[9].float().cpu().permute(0, 3, 1, 2).numpy(), tn.numpy(), **tol)
    d.res = keep[5].data_ptr()   # a residual and a shortcut segment at once is refused
    assert L.lib.mic_bottleneck_fused(C.byref(d), L.stream_ptr()) == -1


def test_bottleneck_fused_rejects_bad_arguments(amd):
    L = amd._lib
    d = L.FusedBlockDesc()
    assert L.lib.mic_bottleneck_fused(C.byref(d), None) == -1
    assert L.lib.mic_bottleneck_fused_supported(L.F32, 64, 64) == 0 and L.lib.mic_bottleneck_fused_supported(L.BF16, 64, 64) == 1
    assert L.lib.mic_bottleneck_fused_supported(L.BF16, 256, 0) == 0 and L.lib.mic_bottleneck_fused_supported(L.BF16, 128, 256) == 0
    t = torch.zeros(1024, dtype=torch.bfloat16, device=DEV)
    for f in ("t1", "w2", "b2", "w3", "b3", "res", "xout"):
        setattr(d, f, t.data_ptr())
    d.dtype, d.B, d.H, d.W, d.C, d.C2 = L.BF16, 1, 2, 2, 64, 0
    d.w2_rows, d.w3_rows = 64, 256   # w2 not padded to 128 rows
    assert L.lib.mic_bottleneck_fused(C.byref(d), None) == -1 and b"shorter than the kernel reads" in L.lib.mic_last_error()
    d.w2_rows = 128
    assert L.lib.mic_bottleneck_fused(C.byref(d), None) == -1 and b"alias" in L.lib.mic_last_error()  # res == xout
