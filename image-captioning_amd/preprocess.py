"""The reference's input transform on the device (runner/config.py:4-12, plotter/captions.py:358-368).

``TRANSFORM = Compose([ToImage(), Resize((256, 256)), ToDtype(float32, scale=True), Normalize(MEAN, STD)])``
takes a decoded RGB image; here the decoded uint8 pixels go to the GPU as they are and one kernel does the
antialiased resize, the scaling and the normalisation (mic_resize_normalize).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Union

import numpy as np
import torch

from . import _lib as L

MEAN = (0.485, 0.456, 0.406)  # runner/config.py:4-5
STD = (0.229, 0.224, 0.225)


class Transform:
    """Callable like the reference's TRANSFORM, for decoded images: a uint8 array / tensor of shape (H, W, 3) or
    (B, H, W, 3) (what ``np.asarray(PIL.Image.open(p).convert("RGB"))`` gives), or a list of (H, W, 3) images of
    different sizes.  Returns fp32 (3, h, w) for one image, (B, 3, h, w) otherwise, on ``device``."""

    def __init__(self, size: Sequence[int] = (256, 256), mean: Sequence[float] = MEAN, std: Sequence[float] = STD,
                 device="cuda", round_to_uint8: bool = True):
        self.size = (int(size[0]), int(size[1]))
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.device = torch.device(device)
        self.round_to_uint8 = bool(round_to_uint8)

    def _one(self, img: torch.Tensor, out: torch.Tensor) -> None:
        if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
            raise ValueError(f"expected uint8 images of shape (B, H, W, 3), got {img.dtype} {tuple(img.shape)}")
        src = img.to(self.device).contiguous()
        B, H, W, _ = src.shape
        L.check(L.lib.mic_resize_normalize(src.data_ptr(), out.data_ptr(), B, H, W, self.size[0], self.size[1], self.mean, self.std,
                                           int(self.round_to_uint8), L.stream_ptr()), "mic_resize_normalize")
        self._keep = src  # the kernel is asynchronous: keep the staging copy until the next call

    def __call__(self, images: Union[np.ndarray, torch.Tensor, list]) -> torch.Tensor:
        if isinstance(images, (list, tuple)):
            out = torch.empty(len(images), 3, *self.size, dtype=torch.float32, device=self.device)
            keep = []
            for i, im in enumerate(images):
                t = torch.as_tensor(np.asarray(im) if not isinstance(im, torch.Tensor) else im)
                self._one(t.unsqueeze(0), out[i:i + 1])
                keep.append(self._keep)
            self._keep = keep
            return out
        t = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        single = t.dim() == 3
        if single:
            t = t.unsqueeze(0)
        out = torch.empty(t.shape[0], 3, *self.size, dtype=torch.float32, device=self.device)
        self._one(t, out)
        return out[0] if single else out
