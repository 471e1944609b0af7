"""ResNet-50 encoders of the three model families and the Swin-V2 encoder on libmic_hip.

Mirrors models/encoders/{transformer,intermediate,basic,swin}.py (forward only; the freeze policy of
models/encoders/base.py is training-side and not part of the generate path).  Output is pixel-major:
the transformer encoder returns image tokens (B, S, H) directly, which is what
``rearrange(features, 'b c h w -> b (h w) c')`` (models/transformer.py:450) produces.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import weights as W


def trunk_out_size(n: int) -> int:
    """Spatial size after the 7x7/2 stem, the 3x3/2 max-pool and the three stride-2 stages."""
    n = (n + 6 - 7) // 2 + 1
    for _ in range(4):
        n = (n + 2 - 3) // 2 + 1
    return n


class ResNet50Encoder:
    """kind: 'transformer' (1x1 conv + ReLU on the 7x7 map), 'intermediate' (avg-pool + Linear + ReLU),
    'basic' (avg-pool + Linear)."""

    def __init__(self, sd: W.SD, kind: str, dtype: torch.dtype, device, prefix: str = "encoder.resnet."):
        if kind not in ("transformer", "intermediate", "basic"):
            raise ValueError(f"Encoder kind {kind} not recognized")
        self.kind, self.dtype, self.device = kind, dtype, torch.device(device)
        self.code = L.dtype_code(dtype)
        self.net, self._keep = W.build_resnet50(sd, prefix, dtype, device)
        if kind == "transformer":
            w, b = sd["encoder.projection.0.weight"], sd["encoder.projection.0.bias"]
            w = w.reshape(w.shape[0], -1)
        elif kind == "intermediate":
            w, b = sd["encoder.linear.0.weight"], sd["encoder.linear.0.bias"]
        else:
            w, b = sd["encoder.linear.weight"], sd["encoder.linear.bias"]
        self.out_dim = w.shape[0]
        self.head_w = W.matrix(w, dtype, device)
        self.head_b = W.vector(b, device)
        self._ws = None

    def _workspace(self, B, H, Wd):
        need = L.lib.mic_resnet50_workspace_bytes(C.byref(self.net), B, H, Wd)
        if need < 0:
            raise ValueError(f"images of {H}x{Wd} are too small for the ResNet-50 trunk")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def trunk(self, images: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) fp32 -> (B, H/32 * W/32, 2048) in the compute dtype."""
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"expected images of shape (B,3,H,W), got {tuple(images.shape)}")
        images = images.to(self.device, torch.float32).contiguous()
        B, _, H, Wd = images.shape
        ws, need = self._workspace(B, H, Wd)
        oh, ow = trunk_out_size(H), trunk_out_size(Wd)
        out = torch.empty(B, oh * ow, 2048, dtype=self.dtype, device=self.device)
        L.check(L.lib.mic_resnet50_forward(C.byref(self.net), images.data_ptr(), B, H, Wd, ws.data_ptr(), need,
                                           out.data_ptr(), L.stream_ptr()), "mic_resnet50_forward")
        return out

    def _linear(self, x: torch.Tensor, rows: int, relu: bool) -> torch.Tensor:
        out = torch.empty(rows, self.out_dim, dtype=self.dtype, device=self.device)
        d = L.ConvDesc()
        d.in_, d.w, d.bias, d.out = x.data_ptr(), self.head_w.data_ptr(), self.head_b.data_ptr(), out.data_ptr()
        d.dtype = d.out_dtype = d.res_dtype = self.code
        d.M, d.N, d.K = rows, self.out_dim, 2048
        d.OH = d.OW = d.IH = d.IW = 1
        d.in_img_stride, d.in_row_stride, d.in_pix_stride = 2048, 0, 4
        d.kw_taps, d.run, d.stride = 1, 2048, 1
        d.ldc, d.relu, d.w_rows = self.out_dim, int(relu), self.head_w.shape[0]
        L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()), "mic_conv2d(encoder head)")
        return out

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        t = self.trunk(images)
        B, S, _ = t.shape
        if self.kind == "transformer":  # models/encoders/transformer.py:46-47
            return self._linear(t, B * S, True).view(B, S, self.out_dim)
        pooled = torch.empty(B, 2048, dtype=self.dtype, device=self.device)
        L.check(L.lib.mic_global_avgpool(t.data_ptr(), pooled.data_ptr(), self.code, B, S, 2048, 2048, L.stream_ptr()),
                "mic_global_avgpool")
        # models/encoders/intermediate.py:47-49 (ReLU) / models/encoders/basic.py:43-45 (no activation)
        return self._linear(pooled, B, self.kind == "intermediate")


class SwinV2Encoder:
    """models/encoders/swin.py: swin_v2_*.features (channels-last, no final norm) -> 1x1 conv + ReLU.
    Returns image tokens (B, S, H), S = ceil(h/32) * ceil(w/32)."""

    def __init__(self, sd: W.SD, dtype: torch.dtype, device, prefix: str = "encoder.features."):
        self.dtype, self.device = dtype, torch.device(device)
        self.code = L.dtype_code(dtype)
        self.net, self._keep = W.build_swin(sd, prefix, dtype, device)
        self.width = self.net.embed_dim * 8
        w, b = sd["encoder.projection.0.weight"], sd["encoder.projection.0.bias"]
        w = w.reshape(w.shape[0], -1)
        if w.shape[1] != self.width:
            raise ValueError(f"projection expects {w.shape[1]} channels, the Swin trunk produces {self.width}")
        self.out_dim = w.shape[0]
        self.head_w = W.matrix(w, dtype, device)
        self.head_b = W.vector(b, device)
        self._ws = None

    @staticmethod
    def out_size(n: int) -> int:
        n //= 4
        for _ in range(3):
            n = (n + 1) // 2
        return n

    def trunk(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"expected images of shape (B,3,H,W), got {tuple(images.shape)}")
        images = images.to(self.device, torch.float32).contiguous()
        B, _, H, Wd = images.shape
        need = L.lib.mic_swin_workspace_bytes(C.byref(self.net), B, H, Wd)
        if need < 0:
            raise ValueError(f"images of {H}x{Wd} are too small for the Swin trunk")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        S = self.out_size(H) * self.out_size(Wd)
        out = torch.empty(B, S, self.width, dtype=self.dtype, device=self.device)
        L.check(L.lib.mic_swin_forward(C.byref(self.net), images.data_ptr(), B, H, Wd, self._ws.data_ptr(), need,
                                       out.data_ptr(), L.stream_ptr()), "mic_swin_forward")
        return out

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        t = self.trunk(images)
        B, S, K = t.shape
        out = torch.empty(B * S, self.out_dim, dtype=self.dtype, device=self.device)
        d = L.ConvDesc()
        d.in_, d.w, d.bias, d.out = t.data_ptr(), self.head_w.data_ptr(), self.head_b.data_ptr(), out.data_ptr()
        d.dtype = d.out_dtype = d.res_dtype = self.code
        d.M, d.N, d.K = B * S, self.out_dim, K
        d.OH = d.OW = d.IH = d.IW = 1
        d.in_img_stride, d.in_row_stride, d.in_pix_stride = K, 0, 4
        d.kw_taps, d.run, d.stride = 1, K, 1
        d.ldc, d.relu, d.w_rows = self.out_dim, 1, self.head_w.shape[0]
        L.check(L.lib.mic_conv2d(C.byref(d), L.stream_ptr()), "mic_conv2d(swin projection)")  # models/encoders/swin.py:70
        return out.view(B, S, self.out_dim)
