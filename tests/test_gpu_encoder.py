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
