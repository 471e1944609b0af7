"""Host-side logic that needs no GPU: vocabulary, weight layout, BatchNorm folding, sharding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import image_captioning_amd as amd
from image_captioning_amd import distributed as D
from image_captioning_amd import weights as W


def test_vocabulary_mirror():
    v = amd.Vocabulary.synthetic(50)
    assert len(v) == 50 and [v.str_to_idx(t) for t in ("<PAD>", "<SOS>", "<EOS>", "<UNK>")] == [0, 1, 2, 3]
    assert v.str_to_idx("nope") == 3 and v.idx_to_str(999) == "<UNK>"
    assert v.encode_as_words([1, 4, 5, 2, 0, 0]) == "w4 w5"       # ids 0,1,2 dropped
    assert v.encode_as_words([1, 3, 7]) == "<UNK> w7"              # UNK is kept, as in the reference
    v2 = amd.Vocabulary("word")
    v2.load_dict(v.get_state_dict())
    assert len(v2) == 50 and v2.word_counts["w10"] == 11
    with pytest.raises(ValueError):
        amd.Vocabulary("bpe")
    counts = v.counts_by_id()
    assert counts[0] == 0 and counts[1] == 50 and counts[4] == 5


def test_word_captions_are_detokenised_the_treebank_way():
    """dataset/vocabulary.py:113-117 detokenises with NLTK's TreebankWordDetokenizer.  NLTK is not installed, so the known
    answers are the examples of that class's own documentation (tokens as NLTK's word tokenizer produces them), plus what
    those rules do to caption-like token lists."""
    from image_captioning_amd.detok import treebank_detokenize as d
    assert d(["Good", "muffins", "cost", "$", "3.88", "in", "New", "York.", "Please", "buy", "me", "two", "of", "them.", "Thanks", "."]) == \
        "Good muffins cost $3.88 in New York. Please buy me two of them. Thanks."
    assert d(["hello", ",", "i", "ca", "n't", "feel", "my", "feet", "!", "Help", "!", "!"]) == "hello, i can't feel my feet! Help!!"
    assert d(["hello", ",", "i", "ca", "n't", "feel", ";", "my", "feet", "!", "Help", "!", "!", "He", "said", ":", "Help", ",", "help", "?", "!"]) == \
        "hello, i can't feel; my feet! Help!! He said: Help, help?!"
    assert d(["a", "man", "'s", "dog", "is", "riding", "a", "surfboard", "."]) == "a man's dog is riding a surfboard."
    assert d(["a", "man", ".", "a", "dog", "."]) == "a man . a dog."          # only the text's last period is re-attached
    assert d(["``", "hello", "''", ",", "she", "said", "(", "quietly", ")", "."]) == '"hello", she said (quietly).'
    assert d(["i", "can", "not", "wan", "na", "go", "--", "really"]) == "i cannot wanna go--really"
    assert d([]) == "" and d(["dog"]) == "dog"
    v = amd.Vocabulary("word")
    words = ["a", "dog", ",", "running", "."]
    for i, w in enumerate(words):
        v.stoi_dict[w] = 4 + i
        v.itos_dict[4 + i] = w
    assert v.encode_as_words([1, 4, 5, 6, 7, 8, 2, 0]) == "a dog, running."


def test_output_bias_matches_oracle():
    from oracle import captioner_oracle as O
    v = amd.Vocabulary.synthetic(200)
    np.testing.assert_allclose(W.output_bias_from_counts(v.counts_by_id()).numpy(), O.output_bias_from_counts(v.counts_by_id()).numpy(),
                               rtol=1e-6)
    assert W.output_bias_from_counts([0, 0, 0, 0, 0]).tolist() == [-1e9, -1e9, 0.0, -1e9, 0.0]


def test_fold_bn_equals_conv_then_batchnorm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 9, 9, generator=g)
    w = torch.randn(16, 8, 3, 3, generator=g)
    gam, bet = torch.randn(16, generator=g), torch.randn(16, generator=g)
    mean, var = torch.randn(16, generator=g), torch.rand(16, generator=g) + 0.5
    ref = F.batch_norm(F.conv2d(x, w, None, 1, 1), mean, var, gam, bet, False, 0.0, 1e-5)
    wf, bf = W.fold_bn(w, gam, bet, mean, var)
    np.testing.assert_allclose(F.conv2d(x, wf, bf, 1, 1).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_weight_layouts():
    w = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).view(2, 3, 2, 2)
    m = W.conv_matrix(w, torch.float32, "cpu")
    assert m.shape == (128, 16)  # rows padded to 128, K = 2*2*3 = 12 padded to the 16-element fp32 k-tile
    assert m[1, :12].tolist() == w[1].permute(1, 2, 0).reshape(-1).tolist() and m[1, 12:].abs().sum() == 0 and m[2:].abs().sum() == 0
    s = W.stem_matrix(torch.ones(64, 3, 7, 7), torch.bfloat16, "cpu").float().view(128, 7, 8, 4)
    assert s[:64, :, :7, :3].min() == 1 and s[:64, :, 7].abs().sum() == 0 and s[:64, :, :, 3].abs().sum() == 0 and s[64:].abs().sum() == 0
    f = W.frag_matrix(torch.arange(16 * 32, dtype=torch.float32).view(16, 32), torch.bfloat16, "cpu").view(64, 8)
    # lane = row + 16*chunk holds K elements [8*chunk, 8*chunk+8) of that row
    assert f[5 + 16 * 2].float().tolist() == [5 * 32 + 16 + e for e in range(8)]
    with pytest.raises(ValueError):
        W.frag_matrix(torch.zeros(10, 32), torch.bfloat16, "cpu")
    assert W.k_multiple(torch.float32) == 16 and W.k_multiple(torch.bfloat16) == 32


def test_random_states_have_reference_names_and_sizes():
    sd = W.random_transformer_state(3500, 512, 3, 40)
    assert sum(v.numel() for v in sd.values()) == 13_071_276  # parameter count of the reference transformer decoder at these sizes
    assert sd["decoder_layers.2.cross_attention.mha.in_proj_weight"].shape == (1536, 512)
    rs = W.random_resnet50_state()
    assert sum(v.numel() for k, v in rs.items() if "running" not in k) == 23_508_032
    ls = W.random_lstm_state("intermediate", 100, 32, 24, 2)
    assert ls["decoder.init_h.weight"].shape == (48, 32) and ls["decoder.lstm.weight_ih_l1"].shape == (96, 24)


def test_models_reject_cpu_and_missing_weights():
    v = amd.Vocabulary.synthetic(20)
    m = amd.transformer.ImageCaptioner(v, 2, torch.float32, encoder="none")
    with pytest.raises(RuntimeError, match="load_state_dict"):
        m.to("cuda:0")
    with pytest.raises(KeyError):
        m.load_state_dict({"foo": torch.zeros(1)})
    m.load_state_dict(W.random_transformer_state(20, 32, 1, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.to("cpu")
    with pytest.raises(ValueError):
        amd.transformer.ImageCaptioner(v, 2, torch.float32, encoder="vgg")
    with pytest.raises(TypeError):
        amd.lstm.ImageCaptioner(v)
    with pytest.raises(ValueError):
        amd._lib.dtype_code(torch.float16)


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 100_000):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= (n + world - 1) // world
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


def test_checkpoint_loader_follows_the_reference_factory(tmp_path):
    """A file shaped like train.save_checkpoint's (train.py:390-403) comes back as the model class the
    reference's factory (runner/runner.py:263-288) would build; nothing needs a GPU until .to()."""
    import torch
    import image_captioning_amd as amd
    vocab = amd.Vocabulary.synthetic(50)
    sd = amd.weights.random_transformer_state(50, 32, 2, 12, seed=0)
    cfg = {"model": "transformer", "encoder": "resnet50", "num_heads": 4, "hidden_size": 32, "num_layers": 2}
    path = tmp_path / "BEST.pt"
    torch.save({"epoch": 3, "model_state": sd, "optimizer_state": {"state": {}, "param_groups": []}, "scheduler_state": None,
                "best_val_loss": 1.5, "train_loss": 1.2, "epochs_no_improve": 0, "lr": (1e-4, 1e-3), "config": cfg, "use_rl": False}, path)
    import pytest
    with pytest.raises(KeyError, match="ResNet-50"):  # config says resnet50, the file has no encoder weights
        amd.checkpoint.load_checkpoint(str(path), vocab)
    sd.update(amd.weights.random_swin_state(0, 32, (1, 1, 1, 1), (1, 2, 4, 8)))
    sd.update(amd.weights.random_head_state("swin", 32, 0, in_dim=256))
    cfg["encoder"] = "swin"
    torch.save({"model_state": sd, "config": cfg}, path)
    model, config = amd.checkpoint.load_checkpoint(str(path), vocab, torch.float32)
    assert isinstance(model, amd.transformer.ImageCaptioner) and model.encoder_kind == "swin" and model.num_heads == 4
    assert config["model"] == "transformer" and model.max_len == 12
    lsd = amd.weights.random_lstm_state("intermediate", 50, 16, 24, 2)
    lsd.update(amd.weights.random_resnet50_state(0))
    lsd.update(amd.weights.random_head_state("intermediate", 16, 0))
    torch.save({"model_state": lsd, "config": {"model": "intermediate", "encoder": "resnet50"}}, path)
    model, _ = amd.checkpoint.load_checkpoint(str(path), vocab)
    assert isinstance(model, amd.lstm.IntermediateImageCaptioner)
    torch.save({"model_state": lsd, "config": {"model": "intermediate", "encoder": "swin"}}, path)
    with pytest.raises(ValueError, match="swin"):
        amd.checkpoint.load_checkpoint(str(path), vocab)
    torch.save({"weights": lsd}, path)
    with pytest.raises(ValueError, match="not a checkpoint"):
        amd.checkpoint.load_checkpoint(str(path), vocab)
