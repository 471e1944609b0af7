"""Swin-V2 encoder (models/encoders/swin.py) on the HIP library against the CPU oracle.

Parity unpinned at the architecture level: torchvision is not installed and the reference holds no
fixture for this encoder, so the oracle is a restatement of the published Swin-V2 design
(oracle.captioner_oracle.swin_v2_features).  What these tests pin is that the HIP path (window
attention with index-mapped padding / shift / masking, GELU epilogue, patch merging as a 2x2/2
implicit GEMM, residual-post-norm) computes the same function as that restatement.
fp32: 1e-3 of the feature scale.  bf16: relative L2 error below 3e-2 (tiny) / 5e-2 (24 blocks).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = dict(embed=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8))


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    return m


def _sd(amd, out_dim, seed=0, **cfg):
    sd = amd.weights.random_swin_state(seed, **cfg)
    sd.update(amd.weights.random_head_state("swin", out_dim, 1, in_dim=8 * cfg.get("embed", 96)))
    return sd


def test_patchify4(amd):
    L = amd._lib
    img = torch.randn(2, 3, 20, 28, generator=torch.Generator().manual_seed(0))
    d_img = img.to(DEV)
    out = torch.empty(2, 5, 7, 64, device=DEV)
    L.check(L.lib.mic_patchify4(d_img.data_ptr(), out.data_ptr(), L.F32, 2, 20, 28, L.stream_ptr()))
    ref = torch.zeros(2, 5, 7, 4, 4, 4)
    ref[..., :3] = img.view(2, 3, 5, 4, 7, 4).permute(0, 2, 4, 3, 5, 1)
    assert torch.equal(out.cpu(), ref.view(2, 5, 7, 64))


def test_layernorm_add(amd):
    L = amd._lib
    g = torch.Generator().manual_seed(1)
    N, H = 37, 96
    u, res = torch.randn(N, H, generator=g) * 3, torch.randn(N, H, generator=g)
    ga, be = torch.randn(H, generator=g), torch.randn(H, generator=g)
    ref = res + torch.nn.functional.layer_norm(u, (H,), ga, be, 1e-5)
    d_u, d_x, d_g, d_b = u.to(DEV), res.to(DEV), ga.to(DEV), be.to(DEV)
    d_t = torch.empty(N, H, device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.mic_layernorm_add(d_u.data_ptr(), d_g.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), d_x.data_ptr(), d_t.data_ptr(),
                                    H, L.BF16, N, H, L.stream_ptr()))
    np.testing.assert_allclose(d_x.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d_t.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)
    assert L.lib.mic_layernorm_add(d_u.data_ptr(), d_g.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), d_u.data_ptr(), None,
                                   H, L.F32, N, H, L.stream_ptr()) == -1


@pytest.mark.parametrize("H,W,shift", [(16, 16, 0), (16, 16, 4), (14, 14, 4), (7, 7, 4), (8, 24, 4), (12, 20, 4), (28, 28, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention(amd, H, W, shift, dtype):
    """qkv projection on the host, attention on the device, output projection on the host, against the
    oracle's whole shifted-window attention (padding, roll, mask, inverse)."""
    from oracle import captioner_oracle as O
    L = amd._lib
    heads, Cc, B = 2, 64, 3
    g = torch.Generator().manual_seed(H * 100 + W + shift)
    p = "a."
    sd = {p + "qkv.weight": torch.randn(3 * Cc, Cc, generator=g) * 0.2, p + "qkv.bias": torch.randn(3 * Cc, generator=g) * 0.3,
          p + "proj.weight": torch.eye(Cc), p + "proj.bias": torch.zeros(Cc),
          p + "logit_scale": torch.log(10 * torch.ones(heads, 1, 1)) + torch.tensor([[[0.2]], [[3.0]]]),  # second head hits the clamp
          p + "cpb_mlp.0.weight": torch.randn(512, 2, generator=g) * 0.5, p + "cpb_mlp.0.bias": torch.randn(512, generator=g) * 0.02,
          p + "cpb_mlp.2.weight": torch.randn(heads, 512, generator=g) * 0.1}
    x = torch.randn(B, H, W, Cc, generator=g)
    ref = O.swin_window_attention(sd, p, x, heads, shift)
    qb = sd[p + "qkv.bias"].clone()
    qb[Cc:2 * Cc] = 0
    qkv = torch.nn.functional.linear(x.view(-1, Cc), sd[p + "qkv.weight"], qb)
    bias = O.swin_position_bias(sd, p).contiguous()
    ls = torch.clamp(sd[p + "logit_scale"].view(-1), max=float(np.log(100.0))).exp()
    d_qkv, d_qb, d_bias, d_ls = qkv.to(DEV, dtype).contiguous(), qb.to(DEV), bias.to(DEV), ls.to(DEV)
    d_o = torch.full((B * H * W, Cc), float("nan"), device=DEV, dtype=dtype)
    L.check(L.lib.mic_window_attn(d_qkv.data_ptr(), d_qb.data_ptr(), d_bias.data_ptr(), d_ls.data_ptr(), d_o.data_ptr(),
                                  L.dtype_code(dtype), B, H, W, Cc, heads, shift, L.stream_ptr()))
    got = d_o.float().cpu().view(B, H, W, Cc)
    assert torch.isfinite(got).all()
    if dtype == torch.float32:
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=1e-4)
    else:
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 2e-2, f"bf16 window attention relative L2 error {rel:.4f}"


def test_window_attention_rejects_bad_arguments(amd):
    L = amd._lib
    t = torch.zeros(64 * 96, device=DEV)
    args = [t.data_ptr()] * 5
    assert L.lib.mic_window_attn(*args, L.F32, 1, 8, 8, 48, 2, 0, L.stream_ptr()) == -1   # 24 channels per head
    assert L.lib.mic_window_attn(*args, L.F32, 1, 8, 8, 64, 2, 8, L.stream_ptr()) == -1   # shift outside the window
    assert b"heads" in L.lib.mic_last_error() or b"shift" in L.lib.mic_last_error()


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 224, 224), (2, 96, 160), (1, 256, 256), (2, 100, 76)])
def test_swin_tiny_fp32_matches_oracle(amd, B, H, W):
    from oracle import captioner_oracle as O
    sd = _sd(amd, 48, **TINY)
    enc = amd.encoders.SwinV2Encoder(sd, torch.float32, DEV)
    img = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        trunk_ref = O.swin_v2_features(sd, img)
        ref = O.tokens_major(O.encoder_swin(sd, img))
    trunk = enc.trunk(img).float().cpu()
    np.testing.assert_allclose(trunk.numpy(), trunk_ref.reshape(B, -1, trunk_ref.shape[-1]).numpy(), rtol=1e-3,
                               atol=1e-3 * float(trunk_ref.abs().max()))
    got = enc(img).float().cpu()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


def test_swin_s_configuration(amd):
    """The S configuration (96 channels, depths 2/2/18/2, heads 3/6/12/24) at 224x224."""
    from oracle import captioner_oracle as O
    sd = _sd(amd, 64)
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        ref = O.tokens_major(O.encoder_swin(sd, img))
    got = amd.encoders.SwinV2Encoder(sd, torch.float32, DEV)(img).float().cpu()
    assert got.shape == ref.shape == (2, 49, 64)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    got16 = amd.encoders.SwinV2Encoder(sd, torch.bfloat16, DEV)(img).float().cpu()
    rel = float((got16 - ref).norm() / ref.norm())
    assert rel < 5e-2, f"bf16 Swin-S relative L2 error {rel:.4f}"


def test_swin_tiny_bf16_tolerance(amd):
    from oracle import captioner_oracle as O
    sd = _sd(amd, 48, **TINY)
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = O.tokens_major(O.encoder_swin(sd, img))
    got = amd.encoders.SwinV2Encoder(sd, torch.bfloat16, DEV)(img).float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 3e-2, f"bf16 Swin relative L2 error {rel:.4f}"


def test_swin_gen_caption_end_to_end_fp32(amd):
    """images -> Swin-V2 -> transformer beam / greedy through `gen_caption`; both registrations of the trunk
    (encoder.features.* and encoder.swin.features.*, models/encoders/swin.py:26,29) load."""
    from oracle import captioner_oracle as O
    V, Hh, Ly, heads, PT = 300, 64, 2, 2, 20
    sd = _sd(amd, Hh, **TINY)
    sd.update(amd.weights.random_transformer_state(V, Hh, Ly, PT, seed=3))
    vocab = amd.Vocabulary.synthetic(V)
    img = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        feats = O.encoder_swin(sd, img)
    renamed = {(k.replace("encoder.features.", "encoder.swin.features.") if k.startswith("encoder.features.") else k): v
               for k, v in sd.items()}
    for state in (sd, renamed):
        model = amd.transformer.ImageCaptioner.from_state_dict(state, vocab, heads, torch.float32, DEV)
        assert model.encoder_kind == "swin"
        for K in (1, 3):
            ref_tok, ref_lp, _ = O.transformer_generate(sd, heads, feats, 12, None, K)
            caps, lp = amd.gen_caption(model, img, vocab, 12, torch.device(DEV), None, K)
            assert caps == [vocab.encode_as_words(t) for t in ref_tok]
            np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), rtol=1e-3, atol=1e-3)
