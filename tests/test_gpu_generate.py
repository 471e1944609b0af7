"""End-to-end parity of the HIP generate path (through the C-ABI) against

* the golden vectors produced by the real reference (tests/golden/*.npz), and
* the CPU oracle on seeded inputs,

for the transformer and LSTM families.  fp32 mode must reproduce token ids exactly and
log-probs / attention within 1e-3 relative (the tolerance the task states for fp32 logits);
bf16 mode is checked at a looser, stated tolerance where a float is compared at all.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32_TOL = dict(rtol=1e-3, atol=1e-4)


def _sd(g, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _rows(arr):
    return [[int(v) for v in r if v >= 0] for r in arr]


class IdVocab:
    """Vocabulary whose 'caption' is the id list itself, so tests compare ids."""

    def __init__(self, size):
        self.size = size

    def str_to_idx(self, w):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[w]

    def encode_as_words(self, ids):
        return [int(i) for i in ids]

    def __len__(self):
        return self.size


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def _tr_model(amd, g, variant, dtype=torch.float32):
    sd = _sd(g)
    sd["output_layer.linear.weight"] = torch.from_numpy(g[f"gen/{variant}/out_weight"])
    sd["output_layer.linear.bias"] = torch.from_numpy(g[f"gen/{variant}/out_bias"])
    V, heads = int(g["cfg"][3]), int(g["cfg"][1])
    return amd.transformer.ImageCaptioner.from_state_dict(sd, IdVocab(V), heads, dtype, "cuda:0")


@pytest.mark.parametrize("variant", ["plain", "eos", "eos_late", "tie"])
@pytest.mark.parametrize("S", [49, 64])
@pytest.mark.parametrize("T", [12, 30])
def test_transformer_golden_fp32(amd, golden_transformer, variant, S, T):
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, variant)
    feats = O.tokens_major(torch.from_numpy(g[f"gen/{variant}/S{S}/features"]))
    key = f"gen/{variant}/S{S}/T{T}"
    vocab = model.vocab
    caps, lp, attn = model.generate_from_features(feats, vocab, T, None, 1, True)
    assert caps == _rows(g[f"{key}/greedy/tokens"])
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{key}/greedy/logp"], **F32_TOL)
    np.testing.assert_allclose(np.array(attn, dtype=np.float32), g[f"{key}/greedy/attn"], **F32_TOL)
    for K in (3, 5):
        caps, lp, attn = model.generate_from_features(feats, vocab, T, None, K, True)
        assert caps == _rows(g[f"{key}/beam{K}/tokens"]), f"beam {K}"
        np.testing.assert_allclose(lp.cpu().numpy(), g[f"{key}/beam{K}/logp"], **F32_TOL)
        ref = g[f"{key}/beam{K}/attn"]
        for i, a in enumerate(attn):
            assert len(a) == len(caps[i]) - 1
            np.testing.assert_allclose(np.array(a, dtype=np.float32), ref[i, : len(a)], **F32_TOL)


def test_transformer_golden_without_attn_and_sync_every_step(amd, golden_transformer):
    """Same tokens when attention is not collected and when the host checks for completion after
    every step (early exit) instead of every 8."""
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, "eos")
    model.sync_every = 1
    feats = O.tokens_major(torch.from_numpy(g["gen/eos/S49/features"]))
    out = model.generate_from_features(feats, model.vocab, 30, None, 1)
    assert len(out) == 2 and out[0] == _rows(g["gen/eos/S49/T30/greedy/tokens"])
    out = model.generate_from_features(feats, model.vocab, 30, None, 3)
    assert len(out) == 2 and out[0] == _rows(g["gen/eos/S49/T30/beam3/tokens"])


@pytest.mark.parametrize("groups", [2, 3, 7])
def test_transformer_beam_in_concurrent_sub_batches(amd, golden_transformer, groups):
    """Beam search split into sub-batches on side streams returns exactly what one batch returns (golden
    ids, scores, attention), also when there are more groups than images."""
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, "eos_late")
    model.decode_groups = groups
    feats = O.tokens_major(torch.from_numpy(g["gen/eos_late/S49/features"]))
    for K in (3, 5):
        caps, lp, attn = model.generate_from_features(feats, model.vocab, 30, None, K, True)
        assert caps == _rows(g[f"gen/eos_late/S49/T30/beam{K}/tokens"])
        np.testing.assert_allclose(lp.cpu().numpy(), g[f"gen/eos_late/S49/T30/beam{K}/logp"], **F32_TOL)
        ref = g[f"gen/eos_late/S49/T30/beam{K}/attn"]
        for i, a in enumerate(attn):
            np.testing.assert_allclose(np.array(a, dtype=np.float32), ref[i, : len(a)], **F32_TOL)


@pytest.mark.parametrize("S", [49, 64])
def test_transformer_teacher_forced_logits_fp32(amd, golden_transformer, S):
    """Stepping the cached decoder over a fixed caption reproduces the reference's full-prefix logits."""
    import ctypes as C
    g = golden_transformer
    from oracle import captioner_oracle as O
    model = _tr_model(amd, g, "plain")
    L = amd._lib
    feats = O.tokens_major(torch.from_numpy(g[f"tf{S}/features"])).to("cuda:0").contiguous()
    cap = torch.from_numpy(g[f"tf{S}/captions"]).to(torch.int32).to("cuda:0").contiguous()
    B, T = cap.shape
    ws, t = model._workspace(B, T, S, 1, 1, 0)
    L.check(L.lib.mic_tr_cross_kv(C.byref(model.dec), C.byref(ws), feats.data_ptr(), L.stream_ptr()))
    got = []
    for p in range(T):
        L.check(L.lib.mic_tr_step(C.byref(model.dec), C.byref(ws), cap.data_ptr(), T, None, p, 0, L.stream_ptr()))
        got.append(t["logits"].cpu().clone())
    got = torch.stack(got, dim=1).numpy()
    # the reference's untied logits of the tiny model are O(1); banned ids sit at -1e9
    ref = torch.from_numpy(g[f"tf{S}/logits"]).numpy()
    live = np.abs(ref) < 1e8
    np.testing.assert_allclose(got[live], ref[live], rtol=1e-3, atol=1e-3)


def test_transformer_bf16_tracks_fp32(amd, golden_transformer):
    """bf16 weights / GEMM inputs: greedy tokens may legitimately differ where two logits are closer
    than bf16 resolution, so the check is on the summed log-prob and on the agreement rate.  Measured on the
    MI355X: every token of the four golden captions agrees and the beam scores differ by 2e-4 relative; the
    bounds leave room for a few near-tie flips (10 % of the tokens) and 25x the score error."""
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, "plain", torch.bfloat16)
    feats = O.tokens_major(torch.from_numpy(g["gen/plain/S49/features"]))
    caps, lp = model.generate_from_features(feats, model.vocab, 12, None, 1)
    ref = _rows(g["gen/plain/S49/T12/greedy/tokens"])
    agree = np.mean([a == b for r, c in zip(ref, caps) for a, b in zip(r, c)])
    assert agree >= 0.9, f"bf16 greedy agreement {agree:.2f}"
    caps, lp = model.generate_from_features(feats, model.vocab, 12, None, 3)
    np.testing.assert_allclose(lp.cpu().numpy(), g["gen/plain/S49/T12/beam3/logp"], rtol=5e-3)


def test_transformer_sampling_matches_oracle_on_shared_uniforms(amd, golden_transformer):
    """Temperature sampling with caller-supplied uniforms equals the oracle's inverse-CDF sampler.
    Parity with torch.multinomial's own random stream is unpinned by construction."""
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, "eos_late")
    sd = _sd(g)
    sd["output_layer.linear.weight"] = torch.from_numpy(g["gen/eos_late/out_weight"])
    sd["output_layer.linear.bias"] = torch.from_numpy(g["gen/eos_late/out_bias"])
    fb = torch.from_numpy(g["gen/eos_late/S49/features"])
    u = torch.rand(16, 4, generator=torch.Generator().manual_seed(11))
    ref_tok, ref_lp, _ = O.transformer_generate(sd, int(g["cfg"][1]), fb, 14, 0.8, 1, O.inverse_cdf_sampler(u.double()))
    caps, lp = model.generate_from_features(O.tokens_major(fb), model.vocab, 14, 0.8, 1, uniforms=u)
    assert caps == ref_tok
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), **F32_TOL)


# ------------------------------------------------------------------------------------------------
# LSTM family
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["plain", "eos"])
@pytest.mark.parametrize("T", [10, 20])
def test_lstm_intermediate_golden_fp32(amd, golden_lstm, variant, T):
    g = golden_lstm
    sd = _sd(g, "inter/sd/")
    sd["decoder.linear.bias"] = torch.from_numpy(g[f"inter/gen/{variant}/out_bias"])
    V = int(g["cfg"][3])
    model = amd.lstm.IntermediateImageCaptioner.from_state_dict(sd, IdVocab(V), torch.float32, "cuda:0")
    feats = torch.from_numpy(g["inter/fwd/features"])
    key = f"inter/gen/{variant}/T{T}"
    assert model.generate_from_features(feats, None, T, None, 1) == _rows(g[f"{key}/greedy/tokens"])
    for K in (3, 5):
        assert model.generate_from_features(feats, None, T, None, K) == _rows(g[f"{key}/beam{K}/tokens"]), f"beam {K}"


def test_lstm_basic_golden_fp32(amd, golden_lstm):
    g = golden_lstm
    sd = _sd(g, "basic/sd/")
    model = amd.lstm.BasicImageCaptioner.from_state_dict(sd, IdVocab(int(g["cfg"][3])), torch.float32, "cuda:0")
    feats = torch.from_numpy(g["inter/fwd/features"])
    assert model.generate_from_features(feats, None, 10, None, 1) == _rows(g["basic/gen/plain/T10/greedy/tokens"])
    assert model.generate_from_features(feats, None, 10, None, 3) == _rows(g["basic/gen/plain/T10/beam3/tokens"])


def test_lstm_sampling_matches_oracle_on_shared_uniforms(amd, golden_lstm):
    from oracle import captioner_oracle as O
    g = golden_lstm
    sd = _sd(g, "inter/sd/")
    model = amd.lstm.IntermediateImageCaptioner.from_state_dict(sd, IdVocab(int(g["cfg"][3])), torch.float32, "cuda:0")
    feats = torch.from_numpy(g["inter/fwd/features"])
    u = torch.rand(12, 4, generator=torch.Generator().manual_seed(5))
    ref = O.lstm_generate(sd, "intermediate", feats, 12, 1.3, 1, O.inverse_cdf_sampler(u.double()))
    assert model.generate_from_features(feats, None, 12, 1.3, 1, uniforms=u) == ref


@pytest.mark.parametrize("heads", [1, 2])
def test_transformer_fp32_hidden_256_matches_oracle(amd, heads):
    """A decoder at the hidden size whose fp32 rows are 1 KiB: the shapes that take the wave-per-position attention
    kernels and, with one or two heads, the cross-attention sub-layer with folded projections.  Not covered by the
    reference-generated fixtures (H=64), so the check is against the oracle running the same random weights."""
    from oracle import captioner_oracle as O
    V, H, Ly, PT, B, S = 400, 256, 2, 24, 5, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=11 + heads)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):  # biases and LayerNorm parameters away from their trivial initial values
        if k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    vocab = IdVocab(V)
    model = amd.transformer.ImageCaptioner.from_state_dict(sd, vocab, heads, torch.float32, "cuda:0")
    feats_bchw = torch.randn(B, H, 7, 7, generator=g).relu()
    feats = O.tokens_major(feats_bchw)
    for K in (1, 3, 5):
        ref_tok, ref_lp, ref_attn = O.transformer_generate(sd, heads, feats_bchw, 14, None, K)
        caps, lp, attn = model.generate_from_features(feats, vocab, 14, None, K, True)
        assert caps == [vocab.encode_as_words(t) for t in ref_tok], f"beam {K}"
        np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), **F32_TOL)
        if K > 1:
            for i, a in enumerate(attn):
                np.testing.assert_allclose(np.array(a, dtype=np.float32), np.array(ref_attn[i], dtype=np.float32)[: len(a)], **F32_TOL)


@pytest.mark.parametrize("S", [49, 64])
def test_transformer_forward_and_evaluation_loss(amd, golden_transformer, S):
    """The public teacher-forced forward (models/transformer.py:338-363) against the reference's logits, and the
    evaluation loss of train.forward_pass (train.py:318-354) against cross-entropy of those reference logits."""
    import torch.nn.functional as F
    from oracle import captioner_oracle as O
    g = golden_transformer
    model = _tr_model(amd, g, "plain")
    feats = O.tokens_major(torch.from_numpy(g[f"tf{S}/features"]))
    cap = torch.from_numpy(g[f"tf{S}/captions"]).long()
    ref = torch.from_numpy(g[f"tf{S}/logits"])
    got = model.forward_from_features(feats, cap).cpu()
    live = ref.abs() < 1e8
    np.testing.assert_allclose(got[live].numpy(), ref[live].numpy(), rtol=1e-3, atol=1e-3)
    # loss: feed captions[:, :-1], score captions[:, 1:], skipping PAD / SOS / UNK targets
    loss, n = model.evaluation_loss(None, cap, model.vocab, feats=feats)
    tgt = cap[:, 1:]
    mask = (tgt != 0) & (tgt != 1) & (tgt != 3)
    ce = F.cross_entropy(ref[:, :-1].reshape(-1, ref.shape[-1]), tgt.reshape(-1), reduction="none").view(tgt.shape)
    assert n == int(mask.sum())
    np.testing.assert_allclose(float(loss), float((ce * mask).sum()), rtol=1e-3)
    with pytest.raises(ValueError):
        model.forward_from_features(feats, cap[:1])
