"""The one-launch decoder step (mic_tr_step_persistent: clusters of 16 workgroups that hand rows to each other inside the
launch) against mic_tr_step (one launch per sub-layer) on the same inputs, through the C-ABI: logits, pre-norm sums,
appended cache rows and attention maps must be identical bit for bit, for one and several rows per image, with ancestry,
with a ragged last cluster, in both dtypes; and whole searches through it must return the ids of the other schedule."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


class IdVocab:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def str_to_idx(self, s):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[s]

    def encode_as_words(self, ids):
        return [int(i) for i in ids]


def _run_steps(L, model, fn_name, B, R, S, T, feats, caps, anc, attn_mode):
    N = B * R
    ws, t = model._workspace(N, T, S, R, max(R, 1), attn_mode)
    assert "sync" in t, "the workspace did not get seam counters: shape not supported by the one-launch step?"
    for k in ("vc", "y", "stats"):
        t[k].zero_()
    for k in ("kc", "vfc"):  # cache cells no step has written hold NaN: both schedules must select masked positions away
        t[k].fill_(float("nan"))
    if attn_mode:
        t["attn"].zero_()
    L.check(L.lib.mic_tr_cross_kv(C.byref(model.dec), C.byref(ws), feats.data_ptr(), L.stream_ptr()), "mic_tr_cross_kv")
    fn = getattr(L.lib, fn_name)
    logits = []
    for p in range(T):
        L.check(fn(C.byref(model.dec), C.byref(ws), caps.data_ptr(), T, L.ptr(anc), p, attn_mode, L.stream_ptr()), fn_name)
        logits.append(t["logits"].clone())
    torch.cuda.synchronize()
    model.check_errors()
    out = [torch.stack(logits), t["kc"].clone(), t["vfc"].clone(), t["y"].clone()]
    if attn_mode:
        out.append(t["attn"].clone())
    return out


@pytest.mark.parametrize("dtype,H,heads,B,R,T", [
    (torch.bfloat16, 512, 2, 256, 3, 6),   # the benchmark's decode shape: 16 full clusters
    (torch.bfloat16, 512, 2, 40, 3, 6),    # ragged: the third cluster holds 8 images
    (torch.bfloat16, 512, 2, 21, 1, 6),    # one row per image (greedy / sampling)
    (torch.bfloat16, 512, 1, 17, 2, 6),
    (torch.float32, 256, 2, 33, 3, 6),
    (torch.float32, 256, 1, 16, 1, 6),
    (torch.bfloat16, 512, 2, 20, 3, 41),   # 41 cached positions (the reference's default max_caption_len = 40): six positions per wave
    (torch.float32, 256, 2, 9, 2, 41),
    (torch.bfloat16, 512, 1, 5, 1, 48),    # the most the fused attention takes
])
def test_one_launch_step_equals_launch_per_sublayer_step(amd, dtype, H, heads, B, R, T):
    L = amd._lib
    V, Ly, PT, S = 500, 3, 48, 49
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=5)
    model = amd.transformer.ImageCaptioner(IdVocab(V), heads, dtype, encoder="none").load_state_dict(sd).to(DEV)
    g = torch.Generator().manual_seed(23)
    N = B * R
    feats = torch.randn(B, S, H, generator=g).relu().to(DEV, dtype).contiguous()
    caps = torch.randint(4, V, (N, T), generator=g, dtype=torch.int32).to(DEV)
    anc = None
    if R > 1:  # every cached position of a row comes from some row of the same image
        img = torch.arange(N) // R
        anc = (img[:, None] * R + torch.randint(0, R, (N, T), generator=g)).to(torch.int32).to(DEV)
    for attn_mode in ((0, 1) if R > 1 else (0, 2)):
        ws, _ = model._workspace(N, T, S, R, max(R, 1), attn_mode)
        assert model.step_schedule(ws) == "persistent"
        ref = _run_steps(L, model, "mic_tr_step", B, R, S, T, feats, caps, anc, attn_mode)
        got = _run_steps(L, model, "mic_tr_step_persistent", B, R, S, T, feats, caps, anc, attn_mode)
        names = ["logits", "kcache", "vfcache", "y", "attn"]
        for name, a, b in zip(names, ref, got):
            assert torch.equal(a, b), f"{name} differs (attn_mode {attn_mode}): max |d| = {(a.float() - b.float()).abs().max().item()}"
        assert bool(torch.isfinite(got[0]).all()) and float(got[0].abs().max()) > 0


@pytest.mark.parametrize("dtype,H", [(torch.bfloat16, 512), (torch.float32, 256)])
def test_searches_through_the_one_launch_step_return_the_same_ids(amd, dtype, H):
    V, Ly, PT, S, B = 500, 3, 31, 49, 48
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=9)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(B, S, H, generator=g).relu()
    res = {}
    for persistent in (False, True):
        model = amd.transformer.ImageCaptioner(IdVocab(V), 2, dtype, encoder="none", persistent=persistent).load_state_dict(sd).to(DEV)
        tok, ln, sc, attn = model.beam_search_ids(feats, 20, 3, return_attn=True)
        sched_beam = model.last_schedule
        gt, gl, _ = model.sample_ids(feats, 20, None)
        res[persistent] = (tok.cpu(), ln.cpu(), sc.cpu(), attn.cpu(), gt.cpu(), gl.cpu())
        model.check_errors()
        assert sched_beam == ("persistent" if persistent else "fused14"), sched_beam
        assert model.last_schedule == ("persistent" if persistent else "fused14"), model.last_schedule
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_one_launch_step_refuses_unsupported_shapes(amd):
    L = amd._lib
    assert L.lib.mic_tr_persist_supported(1, 512, 2, 3, 49, 3, 31, 768) == 1
    assert L.lib.mic_tr_persist_supported(1, 512, 2, 3, 49, 5, 31, 1280) == 0    # five rows per image
    assert L.lib.mic_tr_persist_supported(1, 512, 2, 3, 49, 3, 41, 768) == 1     # the reference's default max_caption_len
    assert L.lib.mic_tr_persist_supported(1, 512, 2, 3, 49, 3, 49, 768) == 0     # more than 48 cached positions
    assert L.lib.mic_tr_persist_supported(1, 512, 2, 3, 49, 3, 31, 3 * 300) == 0  # more images than 16 clusters hold
    assert L.lib.mic_tr_persist_supported(1, 256, 2, 3, 49, 3, 31, 768) == 0     # bf16 rows of 512 bytes
