"""ResNet-50 encoders and the whole generate call (images in, captions out) against the CPU oracle.

The oracle's trunk is a restatement of torchvision's ResNet-50 v1.5 from torch.nn.functional ops
(torchvision is not installed, so the architecture itself is unpinned); what these tests pin is that
the HIP implicit-GEMM path (BN folded, NHWC, fused bias/ReLU/residual) computes the same function.
fp32: 1e-3 of the feature scale.  bf16 (53 layers of bf16 storage, fp32 accumulation): relative L2
error below 3e-2.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    return m


@pytest.fixture(scope="module")
def trunk_sd():
    from oracle import captioner_oracle as O
    return O.random_resnet50_state(0)


def _full_sd(amd, trunk_sd, kind, out_dim):
    sd = dict(trunk_sd)
    sd.update(amd.weights.random_head_state(kind, out_dim, 1))
    return sd


@pytest.mark.parametrize("kind,B,HW", [("transformer", 2, 64), ("transformer", 1, 224), ("transformer", 3, 96),
                                       ("intermediate", 2, 64), ("basic", 2, 64), ("transformer", 1, 256)])
def test_encoder_fp32_matches_oracle(amd, trunk_sd, kind, B, HW):
    from oracle import captioner_oracle as O
    sd = _full_sd(amd, trunk_sd, kind, 64)
    enc = amd.encoders.ResNet50Encoder(sd, kind, torch.float32, DEV)
    img = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        if kind == "transformer":
            ref = O.tokens_major(O.encoder_transformer(sd, img))
        else:
            ref = (O.encoder_intermediate if kind == "intermediate" else O.encoder_basic)(sd, img)
    got = enc(img).float().cpu()
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * scale)


def test_encoder_non_square_and_odd_width(amd, trunk_sd):
    from oracle import captioner_oracle as O
    sd = _full_sd(amd, trunk_sd, "transformer", 32)
    enc = amd.encoders.ResNet50Encoder(sd, "transformer", torch.float32, DEV)
    img = torch.randn(2, 3, 96, 131, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = O.tokens_major(O.encoder_transformer(sd, img))
    got = enc(img).float().cpu()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    with pytest.raises(ValueError):
        enc(torch.randn(2, 1, 96, 96))


def test_encoder_bf16_tolerance(amd, trunk_sd):
    from oracle import captioner_oracle as O
    sd = _full_sd(amd, trunk_sd, "transformer", 64)
    enc = amd.encoders.ResNet50Encoder(sd, "transformer", torch.bfloat16, DEV)
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        ref = O.tokens_major(O.encoder_transformer(sd, img))
    got = enc(img).float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 3e-2, f"bf16 encoder relative L2 error {rel:.4f}"


def test_gen_caption_end_to_end_fp32(amd, trunk_sd):
    """images -> ResNet-50 -> transformer beam / greedy through `gen_caption`, equal token ids to the
    oracle running the same weights on the CPU (reference-as-written search)."""
    from oracle import captioner_oracle as O
    V, H, Ly, heads, PT = 300, 64, 2, 2, 20
    sd = _full_sd(amd, trunk_sd, "transformer", H)
    sd.update(amd.weights.random_transformer_state(V, H, Ly, PT, seed=3))
    vocab = amd.Vocabulary.synthetic(V)
    model = amd.transformer.ImageCaptioner(vocab, heads, torch.float32).load_state_dict(sd)
    img = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        feats = O.encoder_transformer(sd, img)
    for K in (1, 3):
        ref_tok, ref_lp, _ = O.transformer_generate(sd, heads, feats, 12, None, K)
        caps, lp = amd.gen_caption(model, img, vocab, 12, torch.device(DEV), None, K)
        assert caps == [vocab.encode_as_words(t) for t in ref_tok]
        np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), rtol=1e-3, atol=1e-3)
    # 9 positionals as plotter/captions.py:147 passes them; returns the 3-tuple
    out = amd.gen_caption(model, img, vocab, 12, torch.device(DEV), None, 3, True, True)
    assert len(out) == 3 and len(out[2]) == 3 and out[2][0][0].shape == (4,)
    with pytest.raises(NotImplementedError):
        amd.gen_caption(model, img, vocab, 12, torch.device(DEV), None, 1, False)
    # the reference's misspelt keyword is tolerated by generate()
    assert len(model.generate(images=img, vocab=vocab, max_len=5, device=torch.device(DEV), temp=None, beam_size=1, no_grad=True,
                              return_atnn=True)) == 3


def test_lstm_generate_end_to_end_fp32(amd, trunk_sd):
    from oracle import captioner_oracle as O
    V, E, Hd, Ly = 300, 64, 48, 2
    vocab = amd.Vocabulary.synthetic(V)
    img = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    for kind, cls in (("intermediate", amd.lstm.IntermediateImageCaptioner), ("basic", amd.lstm.BasicImageCaptioner)):
        sd = _full_sd(amd, trunk_sd, kind, E)
        sd.update(amd.weights.random_lstm_state(kind, V, E, Hd, Ly, seed=2))
        sd["decoder.linear.weight"] = sd["decoder.linear.weight"] * 8
        model = cls(vocab, torch.float32).load_state_dict(sd)
        with torch.no_grad():
            feats = (O.encoder_intermediate if kind == "intermediate" else O.encoder_basic)(sd, img)
        for K in (1, 3):
            ref = O.lstm_generate(sd, kind, feats, 9, None, K)
            caps, none = amd.gen_caption(model, img, vocab, 9, torch.device(DEV), None, K)
            assert none is None
            assert caps == [vocab.encode_as_words(t) for t in ref], (kind, K)


@pytest.mark.parametrize("B,H,W", [(2, 224, 224), (3, 96, 160), (1, 65, 71), (2, 64, 64), (1, 250, 33)])
def test_stem_pool_equals_convolution_then_pooling(amd, B, H, W):
    """The one-launch stem (7x7/2 convolution + ReLU + 3x3/2 max-pool from an LDS-resident image patch) against
    mic_conv2d + mic_maxpool3x3s2 on the same packed image: bit for bit, over sizes with partial tiles on both axes,
    odd widths and a map narrower than one tile; and both against torch in fp32 within the bf16 tolerance."""
    import ctypes as C
    import torch.nn.functional as F
    L, Wt = amd._lib, amd.weights
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bias = torch.randn(64, generator=g) * 0.2
    wm = Wt.stem_matrix(w, torch.bfloat16, DEV)
    d_img, d_b = img.to(DEV), bias.to(DEV)
    Hp, Wp = H + 6, W + 8 + (W & 1)
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    pack = torch.empty(B, Hp, Wp, 4, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib.mic_pack_image(d_img.data_ptr(), pack.data_ptr(), L.BF16, B, H, W, 3, 3, 3, 5 + (W & 1), L.stream_ptr()))
    conv = torch.empty(B, H1, W1, 64, dtype=torch.bfloat16, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out = pack.data_ptr(), wm.data_ptr(), d_b.data_ptr(), conv.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = B * H1 * W1, 64, 224, H1, W1, Hp, Wp
    d.in_img_stride, d.in_row_stride, d.in_pix_stride = Hp * Wp * 4, Wp * 4, 4
    d.kw_taps, d.run, d.stride, d.ldc, d.relu, d.w_rows = 1, 32, 2, 64, 1, wm.shape[0]
    L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()))
    two = torch.empty(B, H2, W2, 64, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib.mic_maxpool3x3s2(conv.data_ptr(), two.data_ptr(), L.BF16, B, H1, W1, 64, L.stream_ptr()))
    one = torch.full((B, H2, W2, 64), -1.0, dtype=torch.bfloat16, device=DEV)
    assert L.lib.mic_stem_pool_supported(L.BF16) == 1 and L.lib.mic_stem_pool_supported(L.F32) == 0
    L.check(L.lib.mic_stem_pool(pack.data_ptr(), wm.data_ptr(), d_b.data_ptr(), one.data_ptr(), L.BF16, B, Hp, Wp, H1, W1, wm.shape[0],
                                L.stream_ptr()), "mic_stem_pool")
    torch.cuda.synchronize()
    assert torch.equal(one, two)
    ref = F.max_pool2d(F.relu(F.conv2d(img.bfloat16().float(), w.bfloat16().float(), bias, 2, 3)), 3, 2, 1).permute(0, 2, 3, 1)
    np.testing.assert_allclose(one.float().cpu().numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    # a convolution output that does not fit the packed image is refused
    assert L.lib.mic_stem_pool(pack.data_ptr(), wm.data_ptr(), d_b.data_ptr(), one.data_ptr(), L.BF16, B, Hp, Wp, H1 + 1, W1, wm.shape[0],
                               L.stream_ptr()) == -1
    # with the following bottleneck's 1x1 convolution riding along: the same pooled map, and t1 equal to mic_conv2d on it
    w1 = torch.randn(64, 64, 1, 1, generator=g) * 0.2
    b1 = (torch.randn(64, generator=g) * 0.2).to(DEV)
    w1m = Wt.conv_matrix(w1, torch.bfloat16, DEV)
    t1_two = torch.empty(B, H2, W2, 64, dtype=torch.bfloat16, device=DEV)
    d = L.ConvDesc()
    d.in_, d.w, d.bias, d.out = two.data_ptr(), w1m.data_ptr(), b1.data_ptr(), t1_two.data_ptr()
    d.dtype = d.out_dtype = d.res_dtype = L.BF16
    d.M, d.N, d.K, d.OH, d.OW, d.IH, d.IW = B * H2 * W2, 64, 64, 1, 1, 1, 1
    d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride, d.ldc, d.relu, d.w_rows = 64, 4, 1, 64, 1, 64, 1, w1m.shape[0]
    L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()))
    one2 = torch.full((B, H2, W2, 64), -1.0, dtype=torch.bfloat16, device=DEV)
    t1_one = torch.full((B, H2, W2, 64), -1.0, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib.mic_stem_pool_conv1(pack.data_ptr(), wm.data_ptr(), d_b.data_ptr(), one2.data_ptr(), w1m.data_ptr(), b1.data_ptr(),
                                      t1_one.data_ptr(), L.BF16, B, Hp, Wp, H1, W1, wm.shape[0], w1m.shape[0], L.stream_ptr()),
            "mic_stem_pool_conv1")
    torch.cuda.synchronize()
    assert torch.equal(one2, two) and torch.equal(t1_one, t1_two)
