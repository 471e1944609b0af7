"""On-device input transform against torch's own antialiased bilinear interpolate on the CPU, which is what the
reference's v2.Resize applies to a tensor image (runner/config.py:7-12).  torchvision itself is not installed, so
the chain ToImage -> Resize -> ToDtype(scale) -> Normalize is restated with torch ops here (parity unpinned at
the torchvision boundary).  Tolerances: the float path agrees to 2e-5 of a grey level; the path that rounds the
resized image to uint8 levels agrees with torch's fixed-point uint8 kernel within one level (1/255)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    return m


def _images(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(2, H // 9), max(2, W // 9), generator=g)
    smooth = F.interpolate(low, size=(H, W), mode="bicubic").clamp(0, 1)
    noisy = (smooth * 0.8 + 0.2 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1)
    return (noisy * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()  # (B, H, W, 3)


@pytest.mark.parametrize("H,W", [(375, 500), (256, 256), (100, 80), (1000, 1300), (257, 255), (31, 2000)])
def test_resize_normalize_matches_torch(amd, H, W):
    img = _images(2, H, W, H + W)
    chw = img.permute(0, 3, 1, 2).contiguous()
    ref_f = F.interpolate(chw.float(), size=(256, 256), mode="bilinear", antialias=True)
    got_f = amd.preprocess.Transform(round_to_uint8=False)(img).cpu()
    np.testing.assert_allclose((got_f * STD + MEAN).numpy() * 255, ref_f.numpy(), rtol=0, atol=2e-3)
    ref_u8 = F.interpolate(chw, size=(256, 256), mode="bilinear", antialias=True)  # the reference's uint8 intermediate
    ref = (ref_u8.float() / 255 - MEAN) / STD
    got = amd.preprocess.Transform()(img).cpu()
    assert got.shape == (2, 3, 256, 256)
    levels = ((got - ref) * STD * 255).abs()
    assert float(levels.max()) <= 1.0 + 1e-3, f"max difference {float(levels.max()):.3f} grey levels"
    assert float((levels > 0.5).float().mean()) < 0.35


def test_transform_call_forms(amd):
    t = amd.preprocess.Transform(size=(64, 48))
    a, b = _images(1, 90, 120, 1)[0], _images(1, 70, 50, 2)[0]
    one = t(a)
    assert one.shape == (3, 64, 48)
    both = t([a.numpy(), b])
    assert both.shape == (2, 3, 64, 48)
    assert torch.equal(both[0], one)
    with pytest.raises(ValueError):
        t(torch.zeros(4, 4, 3))
    L = amd._lib
    z = torch.zeros(8, dtype=torch.uint8, device="cuda:0")
    o = torch.zeros(8, device="cuda:0")
    assert L.lib.mic_resize_normalize(z.data_ptr(), o.data_ptr(), 1, 1000, 8, 4, 4, t.mean, t.std, 1, L.stream_ptr()) == -1
