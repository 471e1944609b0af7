"""MI355X-native caption generation behind the generate API of MicaKil/image-captioning.

Public surface (mirrors the reference's names for this path):
  gen_caption                         captioner.py:9-30
  transformer.ImageCaptioner          models/transformer.py:296-640
  lstm.BasicImageCaptioner            models/basic.py:53 + models/image_captioner.py
  lstm.IntermediateImageCaptioner     models/intermediate.py:70 + models/image_captioner.py
  Vocabulary                          dataset/vocabulary.py (generate-side subset)
  preprocess.Transform                runner/config.py:4-12 (TRANSFORM: resize 256x256, scale, normalise)
  checkpoint.load_checkpoint          plotter/captions.py:386-397 (train.py:390-403 files)
All compute goes through csrc/libmic_hip.so (include/mic_hip.h); importing fails if it is not built.
"""
from . import _lib  # noqa: F401  (loads libmic_hip.so or raises)
from .captioner import gen_caption
from .vocabulary import Vocabulary, PAD, SOS, EOS, UNK
from . import transformer, lstm, encoders, weights, checkpoint, preprocess, lanes

__all__ = ["gen_caption", "Vocabulary", "transformer", "lstm", "encoders", "weights", "checkpoint", "preprocess", "lanes", "PAD", "SOS", "EOS", "UNK"]
