"""The CPU restatement in oracle/ against vectors produced by the real reference
(tests/golden/make_golden.py).  Token ids must be identical; floats within 2e-5."""
import numpy as np
import pytest
import torch

from oracle import captioner_oracle as O

TOL = dict(rtol=2e-5, atol=2e-5)


def _sd(g, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _rows(arr):
    return [[int(v) for v in r if v >= 0] for r in arr]


def test_transformer_modules(golden_transformer):
    g = golden_transformer
    sd = _sd(g)
    heads = int(g["cfg"][1])
    x, img, ids = (torch.from_numpy(g["mod/" + k]) for k in ("x", "img", "ids"))
    np.testing.assert_allclose(O.seq_embedding(sd, ids).numpy(), g["mod/seq_embedding"], **TOL)
    np.testing.assert_allclose(O.causal_self_attention(sd, 0, x, heads).numpy(), g["mod/self_attention"], **TOL)
    y, w = O.cross_attention(sd, 0, x, img, heads)
    np.testing.assert_allclose(y.numpy(), g["mod/cross_attention"], **TOL)
    np.testing.assert_allclose(w.numpy(), g["mod/cross_attention_scores"], **TOL)
    np.testing.assert_allclose(O.feed_forward(sd, 0, x).numpy(), g["mod/ff"], **TOL)
    np.testing.assert_allclose(O.decoder_layer(sd, 0, img, x, heads)[0].numpy(), g["mod/decoder_layer"], **TOL)
    np.testing.assert_allclose(O.output_layer(sd, x).numpy(), g["mod/output_layer"], **TOL)


def test_output_bias_adapt(golden_transformer):
    g = golden_transformer
    V = int(g["cfg"][3])
    counts = [0, 50, 50, 0] + [(i % 97) + 1 for i in range(4, V)]
    np.testing.assert_allclose(O.output_bias_from_counts(counts).numpy(), g["sd/output_layer.linear.bias"], rtol=1e-6)


@pytest.mark.parametrize("S", [49, 64])
def test_transformer_teacher_forced_logits(golden_transformer, S):
    g = golden_transformer
    sd = _sd(g)
    img = O.tokens_major(torch.from_numpy(g[f"tf{S}/features"]))
    logits, _ = O.transformer_decode(sd, int(g["cfg"][1]), img, torch.from_numpy(g[f"tf{S}/captions"]))
    np.testing.assert_allclose(logits.numpy(), g[f"tf{S}/logits"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("variant", ["plain", "eos", "eos_late", "tie"])
@pytest.mark.parametrize("S", [49, 64])
@pytest.mark.parametrize("T", [12, 30])
def test_transformer_generate(golden_transformer, variant, S, T):
    g = golden_transformer
    sd = _sd(g)
    sd["output_layer.linear.weight"] = torch.from_numpy(g[f"gen/{variant}/out_weight"])
    sd["output_layer.linear.bias"] = torch.from_numpy(g[f"gen/{variant}/out_bias"])
    heads = int(g["cfg"][1])
    feats = torch.from_numpy(g[f"gen/{variant}/S{S}/features"])
    key = f"gen/{variant}/S{S}/T{T}"
    toks, lp, attn = O.transformer_generate(sd, heads, feats, T, None, 1)
    assert toks == _rows(g[f"{key}/greedy/tokens"])
    np.testing.assert_allclose(lp.numpy(), g[f"{key}/greedy/logp"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(np.array(attn, dtype=np.float32), g[f"{key}/greedy/attn"], **TOL)
    for K in (3, 5):
        toks, lp, attn = O.transformer_generate(sd, heads, feats, T, None, K)
        assert toks == _rows(g[f"{key}/beam{K}/tokens"])
        np.testing.assert_allclose(lp.numpy(), g[f"{key}/beam{K}/logp"], rtol=1e-5, atol=1e-4)
        ref = g[f"{key}/beam{K}/attn"]
        for i, a in enumerate(attn):
            np.testing.assert_allclose(np.array(a, dtype=np.float32), ref[i, : len(a)], **TOL)
            assert np.isnan(ref[i, len(a):]).all()


def test_lstm_decoder_forward(golden_lstm):
    g = golden_lstm
    feats, cap = torch.from_numpy(g["inter/fwd/features"]), torch.from_numpy(g["inter/fwd/captions"])
    sd = _sd(g, "inter/sd/")
    np.testing.assert_allclose(O.intermediate_decoder(sd, feats, cap).numpy(), g["inter/fwd/logits"], rtol=1e-4, atol=1e-4)
    sd = _sd(g, "basic/sd/")
    got = O.basic_decoder(sd, feats, cap).numpy()
    ref = g["basic/fwd/logits"]
    live = np.abs(ref) < 1e8  # banned ids sit at -1e9, where fp32 spacing is 64
    np.testing.assert_allclose(got[live], ref[live], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("variant", ["plain", "eos"])
@pytest.mark.parametrize("T", [10, 20])
def test_lstm_intermediate_generate(golden_lstm, variant, T):
    g = golden_lstm
    sd = _sd(g, "inter/sd/")
    sd["decoder.linear.bias"] = torch.from_numpy(g[f"inter/gen/{variant}/out_bias"])
    feats = torch.from_numpy(g["inter/fwd/features"])
    key = f"inter/gen/{variant}/T{T}"
    assert O.lstm_generate(sd, "intermediate", feats, T, None, 1) == _rows(g[f"{key}/greedy/tokens"])
    for K in (3, 5):
        assert O.lstm_generate(sd, "intermediate", feats, T, None, K) == _rows(g[f"{key}/beam{K}/tokens"])


def test_lstm_basic_generate(golden_lstm):
    g = golden_lstm
    sd = _sd(g, "basic/sd/")
    feats = torch.from_numpy(g["inter/fwd/features"])
    assert O.lstm_generate(sd, "basic", feats, 10, None, 1) == _rows(g["basic/gen/plain/T10/greedy/tokens"])
    assert O.lstm_generate(sd, "basic", feats, 10, None, 3) == _rows(g["basic/gen/plain/T10/beam3/tokens"])


def test_resnet50_shapes():
    sd = O.random_resnet50_state(0)
    sd["encoder.projection.0.weight"] = torch.randn(32, 2048, 1, 1) * 0.02
    sd["encoder.projection.0.bias"] = torch.zeros(32)
    with torch.no_grad():
        y = O.encoder_transformer(sd, torch.randn(1, 3, 64, 64))
    assert y.shape == (1, 32, 2, 2)
    n_params = sum(v.numel() for k, v in sd.items() if "resnet" in k and "running" not in k and "tracked" not in k)
    assert n_params == 23508032  # torchvision resnet50 minus the 2048x1000 fc (+bias)
