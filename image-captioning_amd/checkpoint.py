"""Loader for the reference's training checkpoints.

train.py:390-403 saves ``{'epoch', 'model_state', 'optimizer_state', 'scheduler_state', 'best_val_loss',
'train_loss', 'epochs_no_improve', 'lr', 'config': dict(config), 'use_rl'}`` with ``torch.save``;
plotter/captions.py:386-397 reads it back and rebuilds the model from ``config`` through
runner/runner.py:250-288.  This module does the same for the HIP-backed captioners.  The file is opened
with ``weights_only=True``: nothing in it is executed.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import lstm, transformer


def model_from_config(config: dict, state_dict, vocab, dtype: torch.dtype = torch.bfloat16, device=None, **kw):
    """The branch table of runner/runner.py:263-288 (config['encoder'] x config['model'])."""
    enc, kind = config.get("encoder", "resnet50"), config.get("model")
    if enc not in ("resnet50", "swin"):
        raise ValueError(f"Encoder {enc} not recognized")
    if kind == "transformer":
        heads = int(config["num_heads"])
        m = transformer.ImageCaptioner(vocab, heads, dtype, enc, **kw).load_state_dict(state_dict)
    elif kind in ("basic", "intermediate"):
        if enc == "swin":
            # runner/runner.py:276-281 pairs the Swin encoder, which returns a (B, H, h, w) map, with decoders
            # that take a (B, E) vector; that combination cannot run in the reference either
            raise ValueError("the swin encoder is only usable with the transformer model")
        cls = lstm.BasicImageCaptioner if kind == "basic" else lstm.IntermediateImageCaptioner
        m = cls(vocab, dtype, **kw).load_state_dict(state_dict)
    else:
        raise ValueError(f"Model {kind} not recognized")
    return m.to(device) if device is not None else m


def load_checkpoint(path: str, vocab, dtype: torch.dtype = torch.bfloat16, device=None, **kw) -> Tuple[object, dict]:
    """(model, config) from a reference checkpoint file.  ``vocab`` is the Vocabulary the run was trained
    with (the reference rebuilds it from the dataset named in the config, plotter/captions.py:340-355;
    datasets are outside this library).  ``device=None`` leaves the weights on the host until ``.to()``."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(ckpt, dict) or "model_state" not in ckpt or "config" not in ckpt:
        raise ValueError(f"{path} is not a checkpoint written by train.save_checkpoint (model_state / config missing)")
    config = dict(ckpt["config"])
    model = model_from_config(config, ckpt["model_state"], vocab, dtype, device, **kw)
    return model, config
