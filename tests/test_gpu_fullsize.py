"""Properties of the generate path at the sizes of the benchmark configurations, which the CPU oracle
cannot reach in test time: the same models, batch sizes, beam widths and dtype as bench.py's workloads.

Images are independent (models/transformer.py:585 searches image by image; models/image_captioner.py:118
likewise), so at any size

* permuting the images of a batch must permute token ids, lengths and scores, bit for bit: a beam that read
  another image's cache slot, ancestry row, key/value block or top-k candidates would break it;
* running the same batch twice must give the same bits (no result depends on launch order or on what an
  earlier call left in a workspace).
"""
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def _images(batch, seed):
    return torch.randn(batch, 3, bench.IMG, bench.IMG, generator=torch.Generator(device=DEV).manual_seed(seed), device=DEV)


def _check_permutation(run, batch, n_tokens):
    images = _images(batch, 99)
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(5)).to(DEV)
    first = [t.clone() for t in run(images)]
    again = [t.clone() for t in run(images)]
    moved = [t.clone() for t in run(images[perm].contiguous())]
    tokens = first[0]
    assert tokens.shape == (batch, n_tokens)
    assert int(tokens.min()) >= -1 and int(tokens.max()) < bench.V
    for a, b, c in zip(first, again, moved):
        assert torch.equal(a, b), "two runs over one batch differ"
        assert torch.equal(a[perm], c), "permuting the images did not permute the results"
    return first


def test_transformer_beam3_batch256_permutation_and_determinism(amd):
    wl = bench.WORKLOADS["transformer"]
    model = amd.transformer.ImageCaptioner(amd.Vocabulary.synthetic(bench.V), bench.HEADS, torch.bfloat16, encoder="resnet50",
                                           sync_every=bench.MAX_LEN).load_state_dict(bench.build_state("transformer")).to(DEV)

    def run(images):
        tokens, lengths, score, _ = model.beam_search_ids(model.encode(images), bench.MAX_LEN, wl["beam"])
        return tokens, lengths, score

    tokens, lengths, score = _check_permutation(run, wl["batch"], bench.MAX_LEN + 1)
    # the images matter: scores are those of distinct searches, so the permutation check is not vacuous
    assert score.unique().numel() > wl["batch"] // 2
    assert int(lengths.min()) >= 2 and int(lengths.max()) <= bench.MAX_LEN + 1
    assert bool((tokens[:, 0] == 1).all())


def test_lstm_beam3_batch64_permutation_and_determinism(amd):
    wl = bench.WORKLOADS["lstm"]
    model = amd.lstm.IntermediateImageCaptioner(amd.Vocabulary.synthetic(bench.V), torch.bfloat16,
                                                sync_every=bench.MAX_LEN).load_state_dict(bench.build_state("lstm")).to(DEV)

    def run(images):
        return model.beam_search_ids(model.encoder(images), bench.MAX_LEN, wl["beam"])

    tokens, lengths = _check_permutation(run, wl["batch"], bench.MAX_LEN + 1)
    assert int(lengths.min()) >= 2 and int(lengths.max()) <= bench.MAX_LEN + 1
    assert bool((tokens[:, 0] == 1).all())


def test_swin_beam5_batch512_permutation_and_determinism(amd):
    wl = bench.WORKLOADS["swin"]
    model = amd.transformer.ImageCaptioner(amd.Vocabulary.synthetic(bench.V), bench.HEADS, torch.bfloat16, encoder="swin",
                                           sync_every=bench.MAX_LEN).load_state_dict(bench.build_state("swin")).to(DEV)

    def run(images):
        tokens, lengths, score, _ = model.beam_search_ids(model.encode(images), bench.MAX_LEN, wl["beam"])
        return tokens, lengths, score

    tokens, lengths, score = _check_permutation(run, wl["batch"], bench.MAX_LEN + 1)
    assert score.unique().numel() > wl["batch"] // 2
    assert bool((tokens[:, 0] == 1).all())
