#!/usr/bin/env python3
"""Generate golden input/output vectors by running the real reference model code.

Run once in the build container (where /root/reference is mounted); the
resulting ``*.npz`` files are committed and are the only thing the tests read.
Nothing here is imported by the product path.

The reference cannot be imported verbatim in this image (nltk / torchvision are
not installed, and models.image_captioner <-> models.basic/intermediate import
each other), so the harness installs three in-memory stand-ins before import:

* ``nltk``: only ``word_tokenize`` / ``TreebankWordDetokenizer`` names, used by
  the reference solely to turn ids into a display string.  Fixtures store ids.
* ``torchvision`` / ``torchvision.models``: only the two ``*_Weights`` names the
  encoder modules import.  No torchvision network is ever constructed; the
  encoder is replaced by a module that returns a seeded feature tensor.
* placeholder ``models.basic`` / ``models.intermediate`` while
  ``models.image_captioner`` is first imported (breaks the import cycle), then
  the real modules are imported over them.

Two more harness-side adaptations, neither of which touches token selection:
a forward hook that gives ``CrossAttention.attention_scores`` the 4-D shape
``transformer.temperature_sampling`` indexes (models/transformer.py:502), and
calling ``model.generate`` directly because ``captioner.gen_caption`` passes a
misspelt keyword (captioner.py:30).
"""
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("MIC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------
# import shims
# --------------------------------------------------------------------------------------
def import_reference():
    sys.path.insert(0, REF)

    nltk = types.ModuleType("nltk")
    nltk.word_tokenize = lambda s: s.split()

    class _Detok:
        def detokenize(self, toks):
            return " ".join(toks)

    nltk.TreebankWordDetokenizer = _Detok
    sys.modules["nltk"] = nltk

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    class _W:
        IMAGENET1K_V1 = None
        IMAGENET1K_V2 = None

    tvm.ResNet50_Weights = _W
    tvm.Swin_V2_S_Weights = _W
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm

    import models  # noqa: F401  (reference package)

    for name in ("models.basic", "models.intermediate"):
        dummy = types.ModuleType(name)
        dummy.Decoder = type("Decoder", (), {})
        sys.modules[name] = dummy
        setattr(sys.modules["models"], name.split(".")[1], dummy)
    import models.image_captioner as ic  # noqa: F401

    for name in ("models.basic", "models.intermediate"):
        del sys.modules[name]
        delattr(sys.modules["models"], name.split(".")[1])
    import models.basic as basic
    import models.intermediate as inter
    import models.transformer as tr
    from dataset.vocabulary import Vocabulary

    return ic, basic, inter, tr, Vocabulary


class FixedEncoder(nn.Module):
    """Stands in for the torchvision-backed encoder: returns a preset tensor."""

    def __init__(self):
        super().__init__()
        self.out = None

    def forward(self, images):
        return self.out


def make_vocab(Vocabulary, V):
    vocab = Vocabulary("word")
    for i in range(4, V):
        w = f"w{i}"
        vocab.stoi_dict[w] = i
        vocab.itos_dict[i] = w
        vocab.word_counts[w] = (i % 97) + 1
    vocab.word_counts["<SOS>"] = 50
    vocab.word_counts["<EOS>"] = 50
    assert len(vocab) == V
    return vocab


class IdRecorder:
    """Wraps vocab.encode_as_words so fixtures keep the id lists, not strings."""

    def __init__(self, vocab):
        self.vocab = vocab
        self.ids = []
        self._orig = vocab.encode_as_words
        vocab.encode_as_words = self

    def __call__(self, idxs):
        self.ids.append([int(i) for i in idxs])
        return self._orig(idxs)

    def take(self):
        out, self.ids = self.ids, []
        return out


def pad_ragged(lists, fill=-1):
    n = max(len(x) for x in lists)
    arr = np.full((len(lists), n), fill, dtype=np.int64)
    for i, x in enumerate(lists):
        arr[i, : len(x)] = x
    return arr


def sd_to_np(sd, prefix="sd/"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# transformer fixtures
# --------------------------------------------------------------------------------------
def transformer_fixtures(tr, Vocabulary):
    H, heads, L, V, PT = 64, 2, 2, 200, 16
    torch.manual_seed(0)
    vocab = make_vocab(Vocabulary, V)
    rec = IdRecorder(vocab)
    enc = FixedEncoder()
    model = tr.ImageCaptioner(enc, vocab, hidden_size=H, num_layers=L, num_heads=heads, max_len=PT, decoder_dropout=0.5)
    model.eval()

    def fix_attn_shape(mod, inp, out):
        mod.attention_scores = mod.attention_scores.unsqueeze(1)

    out = {"cfg": np.array([H, heads, L, V, PT], dtype=np.int64)}
    out.update(sd_to_np({k: v for k, v in model.state_dict().items() if not k.startswith("encoder.")}))

    g = torch.Generator().manual_seed(1)
    # ---- per-module vectors: each sub-module of the encoders and decoders on its own ----
    B, t, S = 3, 5, 49
    ids = torch.randint(4, V, (B, t), generator=g)
    x = torch.randn(B, t, H, generator=g)
    img = torch.randn(B, S, H, generator=g)
    with torch.no_grad():
        layer = model.decoder_layers[0]
        out["mod/ids"] = ids.numpy()
        out["mod/x"] = x.numpy()
        out["mod/img"] = img.numpy()
        out["mod/seq_embedding"] = model.seq_embedding(ids).numpy()
        out["mod/self_attention"] = layer.self_attention(x).numpy()
        out["mod/cross_attention"] = layer.cross_attention(x, img).numpy()
        out["mod/cross_attention_scores"] = layer.cross_attention.attention_scores.numpy()
        out["mod/ff"] = layer.ff(x).numpy()
        out["mod/decoder_layer"] = layer(img, x).numpy()
        out["mod/output_layer"] = model.output_layer(x).numpy()

    # ---- teacher-forced full-prefix logits (pins cached == recompute) ----
    for S_ in (49, 64):
        hw = int(S_ ** 0.5)
        feats = torch.randn(4, H, hw, hw, generator=g)
        enc.out = feats
        cap = torch.randint(4, V, (4, PT), generator=g)
        cap[:, 0] = 1
        with torch.no_grad():
            logits = model(torch.zeros(4, 3, 8, 8), cap)
        out[f"tf{S_}/features"] = feats.numpy()
        out[f"tf{S_}/captions"] = cap.numpy()
        out[f"tf{S_}/logits"] = logits.numpy()

    # ---- generation: variants of the output bias ----
    base_bias = model.output_layer.linear.bias.data.clone()
    base_w = model.output_layer.linear.weight.data.clone()
    variants = {"plain": (base_w, base_bias)}
    b2 = base_bias.clone()
    b2[2] = b2[4:].max() + 0.3  # EOS competitive: greedy and beams finish at different early steps
    variants["eos"] = (base_w, b2)
    b2 = base_bias.clone()
    b2[2] = b2[4:].max() - 0.3  # EOS weaker: beams finish late, at different steps; greedy never does
    variants["eos_late"] = (base_w, b2)
    # bias-only model: every row has the same distribution -> exact fp64 score ties
    gb = torch.Generator().manual_seed(7)
    b3 = torch.randn(V, generator=gb)
    b3[[0, 1, 3]] = -1e9
    b3[2] = b3[4:].topk(3).values[2] - 0.05
    variants["tie"] = (torch.zeros_like(base_w), b3)

    for vname, (w, b) in variants.items():
        model.output_layer.linear.weight.data = w.clone()
        model.output_layer.linear.bias.data = b.clone()
        out[f"gen/{vname}/out_weight"] = w.numpy()
        out[f"gen/{vname}/out_bias"] = b.numpy()
        for S_ in (49, 64):
            hw = int(S_ ** 0.5)
            gB = torch.Generator().manual_seed(100 + S_)
            feats = torch.randn(4, H, hw, hw, generator=gB)
            enc.out = feats
            out[f"gen/{vname}/S{S_}/features"] = feats.numpy()
            images = torch.zeros(4, 3, 8, 8)
            for max_len in (12, 30):  # 30 exceeds the 16-row position table -> clamp path
                key = f"gen/{vname}/S{S_}/T{max_len}"
                # greedy (needs the attention-shape hook)
                hooks = [l.cross_attention.register_forward_hook(fix_attn_shape) for l in model.decoder_layers]
                caps, lp, attn = model.generate(images, vocab, max_len, torch.device("cpu"), None, 1, True, True)
                for h in hooks:
                    h.remove()
                ids_ = rec.take()
                out[f"{key}/greedy/tokens"] = pad_ragged(ids_)
                out[f"{key}/greedy/logp"] = lp.numpy()
                out[f"{key}/greedy/attn"] = np.array(attn, dtype=np.float32)  # (L, steps, B, S)
                for K in (3, 5):
                    caps, lp, attn = model.generate(images, vocab, max_len, torch.device("cpu"), None, K, True, True)
                    ids_ = rec.take()
                    out[f"{key}/beam{K}/tokens"] = pad_ragged(ids_)
                    out[f"{key}/beam{K}/logp"] = lp.numpy()
                    steps = max(len(a) for a in attn)
                    A = np.full((len(attn), steps, S_), np.nan, dtype=np.float32)
                    for i, a in enumerate(attn):
                        A[i, : len(a)] = np.array(a, dtype=np.float32)
                    out[f"{key}/beam{K}/attn"] = A
    np.savez_compressed(os.path.join(OUT, "transformer_tiny.npz"), **out)
    print("transformer_tiny.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------
# LSTM fixtures
# --------------------------------------------------------------------------------------
def lstm_fixtures(basic, inter, Vocabulary):
    E, Hd, L, V = 48, 40, 2, 200
    torch.manual_seed(3)
    vocab = make_vocab(Vocabulary, V)
    rec = IdRecorder(vocab)
    out = {"cfg": np.array([E, Hd, L, V], dtype=np.int64)}

    # ---- intermediate ----
    dec = inter.Decoder(E, Hd, vocab, 0.5, L, 0)
    dec.linear.weight.data *= 6.0  # default init gives near-constant logits; widen them
    enc = FixedEncoder()
    model = inter.ImageCaptioner(enc, dec)
    model.eval()
    out.update(sd_to_np(dec.state_dict(), "inter/sd/decoder."))
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(4, E, generator=g)
    cap = torch.randint(4, V, (4, 9), generator=g)
    with torch.no_grad():
        out["inter/fwd/features"] = feats.numpy()
        out["inter/fwd/captions"] = cap.numpy()
        out["inter/fwd/logits"] = dec(feats, cap).numpy()
    base_b = dec.linear.bias.data.clone()
    for vname in ("plain", "eos"):
        b = base_b.clone()
        if vname == "eos":
            b[2] = 10.0
        dec.linear.bias.data = b
        out[f"inter/gen/{vname}/out_bias"] = b.numpy()
        enc.out = feats
        for max_len in (10, 20):
            key = f"inter/gen/{vname}/T{max_len}"
            images = torch.zeros(4, 3, 8, 8)
            caps, none = model.generate(images, vocab, max_len, None, 1, torch.device("cpu"), True)
            assert none is None
            out[f"{key}/greedy/tokens"] = pad_ragged(rec.take())
            for K in (3, 5):
                model.generate(images, vocab, max_len, None, K, torch.device("cpu"), True)
                out[f"{key}/beam{K}/tokens"] = pad_ragged(rec.take())

    # ---- basic: Decoder.__init__ raises (models/basic.py:32 reads an attribute that is
    # never set), so assemble the instance by hand and pin forward() only ----
    torch.manual_seed(4)
    bdec = basic.Decoder.__new__(basic.Decoder)
    nn.Module.__init__(bdec)
    bdec.embed = nn.Embedding(V, E, padding_idx=0)
    bdec.lstm = nn.LSTM(E, Hd, L, batch_first=True, dropout=0.5)
    bdec.linear = nn.Linear(Hd, V)
    bdec.linear.weight.data *= 6.0
    bdec.linear.bias.data.zero_()
    bdec.linear.bias.data[[0, 1, 3]] = -1e9  # the evident intent of basic.py:32 (cf. intermediate.py:30,43)
    bdec.dropout = nn.Dropout(0.5)
    bdec.eval()
    out.update(sd_to_np(bdec.state_dict(), "basic/sd/decoder."))
    with torch.no_grad():
        out["basic/fwd/features"] = feats.numpy()
        out["basic/fwd/captions"] = cap.numpy()
        out["basic/fwd/logits"] = bdec(feats, cap).numpy()
    # generation through the shared LSTM-family search code with the hand-built decoder
    bmodel = basic.ImageCaptioner(enc, bdec)
    bmodel.eval()
    for max_len in (10,):
        key = f"basic/gen/plain/T{max_len}"
        images = torch.zeros(4, 3, 8, 8)
        bmodel.generate(images, vocab, max_len, None, 1, torch.device("cpu"), True)
        out[f"{key}/greedy/tokens"] = pad_ragged(rec.take())
        bmodel.generate(images, vocab, max_len, None, 3, torch.device("cpu"), True)
        out[f"{key}/beam3/tokens"] = pad_ragged(rec.take())
    np.savez_compressed(os.path.join(OUT, "lstm_tiny.npz"), **out)
    print("lstm_tiny.npz:", len(out), "arrays")


if __name__ == "__main__":
    ic, basic, inter, tr, Vocabulary = import_reference()
    torch.set_num_threads(4)
    transformer_fixtures(tr, Vocabulary)
    lstm_fixtures(basic, inter, Vocabulary)
