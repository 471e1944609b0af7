"""CPU checks of the Swin-V2 restatement in the oracle (no fixture of the reference covers this
encoder, so these pin structure, not values: published parameter count, buffer shapes, and the
window machinery against a direct evaluation where one window covers the map)."""
import math

import torch
import torch.nn.functional as F

from oracle import captioner_oracle as O


def test_swin_s_parameter_count_matches_the_published_model():
    sd = O.random_swin_state(0)
    n = sum(v.numel() for k, v in sd.items() if "relative_" not in k)
    # torchvision swin_v2_s: 49,737,442 parameters including head (768*1000 + 1000) and final norm (2*768)
    assert n == 49_737_442 - 769_000 - 1_536
    assert O.swin_config_of(sd) == (96, [2, 2, 18, 2], [3, 6, 12, 24])


def test_relative_tables():
    table, index = O.swin_relative_tables(8)
    assert table.shape == (1, 15, 15, 2) and index.shape == (64 * 64,)
    assert int(index.min()) == 0 and int(index.max()) == 15 * 15 - 1
    assert abs(float(table.abs().max()) - math.log2(9.0) / 3.0) < 1e-6  # sign(8) * log2(8 + 1) / log2(8) at the far corner
    assert int(index.view(64, 64)[0, 0]) == 7 * 15 + 7  # zero offset sits in the table's centre


def test_single_window_equals_direct_cosine_attention():
    g = torch.Generator().manual_seed(0)
    heads, C = 2, 64
    sd = O.random_swin_state(1, 64, (1, 1, 1, 1), (2, 4, 8, 16))
    p = "encoder.features.1.0.attn."
    x = torch.randn(2, 8, 8, C, generator=g)
    got = O.swin_window_attention(sd, p, x, heads, 4)  # the shift is dropped: one window spans both axes
    b = sd[p + "qkv.bias"].clone()
    b[C:2 * C] = 0
    qkv = F.linear(x.view(2, 64, C), sd[p + "qkv.weight"], b).view(2, 64, 3, heads, 32).permute(2, 0, 3, 1, 4)
    q, k, v = F.normalize(qkv[0], dim=-1), F.normalize(qkv[1], dim=-1), qkv[2]
    a = q @ k.transpose(-1, -2) * sd[p + "logit_scale"].clamp(max=math.log(100.0)).exp() + O.swin_position_bias(sd, p)
    ref = F.linear((a.softmax(-1) @ v).transpose(1, 2).reshape(2, 64, C), sd[p + "proj.weight"], sd[p + "proj.bias"])
    torch.testing.assert_close(got.view(2, 64, C), ref, rtol=1e-5, atol=1e-5)


def test_shifted_windows_only_mix_within_regions():
    """With a shift, a token in the wrapped-around strip must not attend across the seam: zeroing every
    other region's input leaves that strip's attention output (before proj) unchanged."""
    sd = O.random_swin_state(2, 32, (2, 1, 1, 1), (1, 2, 4, 8))
    p = "encoder.features.1.1.attn."
    sd[p + "proj.weight"], sd[p + "proj.bias"] = torch.eye(32), torch.zeros(32)
    sd[p + "logit_scale"] = torch.full((1, 1, 1), 10.0)  # clamped to log(100): sharp, mask -100 still dominates
    x = torch.randn(1, 16, 16, 32, generator=torch.Generator().manual_seed(3))
    full = O.swin_window_attention(sd, p, x, 1, 4)
    x2 = x.clone()
    x2[:, 4:, :, :] = 0  # rows 0..3 roll to the bottom strip of the last window row
    part = O.swin_window_attention(sd, p, x2, 1, 4)
    # not bit-equal: masked logits are -100, not -inf, so a residue of exp(-100 + ...) remains
    torch.testing.assert_close(full[:, :4], part[:, :4], rtol=0, atol=1e-6)


def test_features_shapes():
    sd = O.random_swin_state(0, 32, (2, 2, 2, 2), (1, 2, 4, 8))
    sd.update({"encoder.projection.0.weight": torch.randn(16, 256, 1, 1) * 0.05, "encoder.projection.0.bias": torch.zeros(16)})
    y = O.encoder_swin(sd, torch.randn(2, 3, 100, 76))
    assert y.shape == (2, 16, 4, 3) and bool((y >= 0).all())
