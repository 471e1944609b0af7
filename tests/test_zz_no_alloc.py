"""mic_tr_step_static (the decoder step whose temporaries live in a device array inside the library, for callers
that cannot allocate) against mic_tr_step on the same inputs, through the C-ABI: logits and the appended key/value
cache rows must be identical bit for bit (same kernels, same order), on two calls in a row, the second with fewer
rows, and a step too large for the built-in buffer must be refused."""
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


def _steps(L, model, fn_name, N, S, T, feats, caps, strip):
    """cross-K/V, then T decoder steps over the given tokens; returns the logits of every step and the self-attention
    caches.  `strip` clears the workspace's temporaries' pointers first (the static entry must not need them)."""
    ws, t = model._workspace(N, T, S, 1, 1, 0)
    for k in ("kc", "vc"):
        t[k].zero_()
    L.check(L.lib.mic_tr_cross_kv(C.byref(model.dec), C.byref(ws), feats.data_ptr(), L.stream_ptr()), "mic_tr_cross_kv")
    io = L.TrWorkspace.from_buffer_copy(ws)
    if strip:
        io.x = io.xt = io.y = io.qkv = io.o = io.ffh = None
        io.stats[0] = io.stats[1] = None
    fn = getattr(L.lib, fn_name)
    out = []
    for p in range(T):
        L.check(fn(C.byref(model.dec), C.byref(io), caps.data_ptr(), T, None, p, 0, L.stream_ptr()), fn_name)
        out.append(t["logits"].clone())
    torch.cuda.synchronize()
    return torch.stack(out), t["kc"].clone(), t["vc"].clone()


@pytest.mark.parametrize("dtype,H,heads", [(torch.bfloat16, 512, 2), (torch.float32, 64, 2)])
def test_static_scratch_step_equals_workspace_step(amd, dtype, H, heads):
    L = amd._lib
    V, Ly, PT, S, T = 400, 2, 16, 49, 5
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=31)
    model = amd.transformer.ImageCaptioner(IdVocab(V), heads, dtype, encoder="none").load_state_dict(sd).to(DEV)
    g = torch.Generator().manual_seed(17)
    for N in (96, 40):  # two calls in a row on one GPU's array, the second with a smaller input
        feats = torch.randn(N, S, H, generator=g).relu().to(DEV, dtype).contiguous()
        caps = torch.randint(4, V, (N, T), generator=g, dtype=torch.int32).to(DEV)
        ref = _steps(L, model, "mic_tr_step", N, S, T, feats, caps, strip=False)
        got = _steps(L, model, "mic_tr_step_static", N, S, T, feats, caps, strip=True)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        assert bool(torch.isfinite(got[0]).all()) and float(got[0].abs().max()) > 0


def test_static_scratch_step_rejects_what_does_not_fit(amd):
    L = amd._lib
    V, H, Ly, PT = 400, 512, 1, 16
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=2)
    model = amd.transformer.ImageCaptioner(IdVocab(V), 2, torch.bfloat16, encoder="none").load_state_dict(sd).to(DEV)
    io = L.TrWorkspace()
    io.N, io.t_cap, io.S, io.rows_per_image = 768, 8, 49, 1   # 768 rows x 512 need about 10.6 MB of temporaries
    assert L.lib.mic_tr_step_static(C.byref(model.dec), C.byref(io), None, 8, None, 0, 0, L.stream_ptr()) == -1
    assert b"built-in buffer holds" in L.lib.mic_last_error()
    torch.cuda.synchronize()
