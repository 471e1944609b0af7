"""`gen_caption`: the single entry the reference's callers use (captioner.py:9-30)."""
from __future__ import annotations

from typing import Optional

import torch


def gen_caption(model, images: torch.Tensor, vocab, max_len: int = 30, device: torch.device = torch.device("cpu"),
                temp: Optional[float] = None, beam_size: int = 1, no_grad=True, return_attn=False) -> tuple:
    """Same positional order and defaults as the reference; forwards everything by keyword because
    the two model families order `generate`'s parameters differently (models/image_captioner.py:40-41
    vs models/transformer.py:394-395).  The reference misspells the last keyword (captioner.py:30:
    ``return_atnn``), which makes every call raise; the intended name is used here."""
    model = model.to(device)
    return model.generate(images=images, vocab=vocab, max_len=max_len, device=device, temp=temp, beam_size=beam_size,
                          no_grad=no_grad, return_attn=return_attn)
