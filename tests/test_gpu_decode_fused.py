"""GPU parity tests of the fused decode-step kernels: both attention sub-layers of a decoder layer in one launch
(mic_attn_fused) against a plain PyTorch restatement of models/transformer.py:81-140 on the newest position, the tail
of a beam-search step (mic_beam_tail) against the three launches it replaces, and the step drivers that use them
against the unfused path and the fp32 oracle."""
import ctypes as C
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd():
    import image_captioning_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def _stats(raw):
    """Per 128 columns: mean and centred sum of squares, the form the GEMM tails leave behind."""
    N, H = raw.shape
    r = raw.view(N, H // 128, 128).double()
    mean = r.mean(-1)
    m2 = ((r - mean[..., None]) ** 2).sum(-1)
    return torch.stack([mean, m2], -1).float().contiguous()


def _attn_reference(dt, heads, t, R, x, qkvf, kc, vfc, anc, self_b, g1, b1, kf, vf, c, bo):
    """fp64 restatement; kc / vfc / kf / vf hold values already rounded to the compute dtype."""
    N, H = x.shape
    hd = H // heads
    B = N // R
    S = kf.shape[1]
    rnd = (lambda v: v.to(dt).double())
    q = qkvf[:, :H].double() * (float(torch.sqrt(torch.tensor(1.0 / hd, dtype=torch.float32))))
    knew = rnd(qkvf[:, H:2 * H])
    vnew = [rnd(qkvf[:, (2 + h) * H:(3 + h) * H]) for h in range(heads)]
    y = torch.empty(N, H, dtype=torch.float64)
    attn = torch.empty(N, S, dtype=torch.float64)
    for n in range(N):
        K = torch.stack([kc[anc[n, p], p].double() for p in range(t)] + [knew[n]])          # (t+1, H)
        V = torch.stack([vfc[anc[n, p], p].double() for p in range(t)] + [torch.stack([v[n] for v in vnew])])  # (t+1, heads, H)
        out = x[n].double() + self_b.double()
        for h in range(heads):
            sl = slice(h * hd, (h + 1) * hd)
            w = torch.softmax(K[:, sl] @ q[n, sl], 0)
            out = out + w @ V[:, h]
        u = F.layer_norm(out, (H,), g1.double(), b1.double(), 1e-5)
        b = n // R
        o2 = u + bo.double()
        aw = torch.zeros(S, dtype=torch.float64)
        for h in range(heads):
            w = torch.softmax(kf[b, :, h].double() @ u + c[b, :, h].double(), 0)
            o2 = o2 + w @ vf[b, :, h].double()
            aw += w / heads
        y[n] = o2
        attn[n] = aw
    return y, knew, vnew, attn


@pytest.mark.parametrize("dtype,heads,R,B,S,t,t_cap,view", [
    (torch.bfloat16, 2, 3, 5, 49, 0, 31, False), (torch.bfloat16, 2, 3, 5, 49, 7, 31, True), (torch.bfloat16, 2, 3, 4, 64, 30, 31, True),
    (torch.bfloat16, 2, 5, 3, 49, 12, 32, True), (torch.bfloat16, 1, 1, 7, 64, 9, 16, True), (torch.bfloat16, 2, 8, 2, 33, 5, 12, False),
    (torch.bfloat16, 2, 4, 3, 49, 17, 31, True), (torch.float32, 2, 3, 4, 49, 11, 31, True), (torch.float32, 1, 2, 3, 64, 3, 8, False),
    (torch.float32, 2, 1, 6, 49, 31, 32, True)])
def test_attn_fused_matches_reference(amd, dtype, heads, R, B, S, t, t_cap, view):
    L = amd._lib
    H = 512 if dtype == torch.bfloat16 else 256
    N = B * R
    g = torch.Generator().manual_seed(1000 + t + R)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    qkvf = rn(N, (2 + heads) * H)
    kc = rn(N, t_cap, H).to(dtype)
    vfc = rn(N, t_cap, heads, H).to(dtype)
    # cells at and beyond t hold whatever the allocation held: NaN here.  Masked positions must be selected away, not
    # multiplied by a zero weight (at t = 0 every masked load is redirected to the cell the step itself is writing)
    kc[:, t:] = float("nan")
    vfc[:, t:] = float("nan")
    anc = torch.stack([torch.randint(0, R, (t_cap,), generator=g) + (n // R) * R for n in range(N)]).int()
    raw = rn(N, H, sc=2.0) + 0.5
    gam, bet = 1 + 0.2 * rn(H), 0.2 * rn(H)
    x = F.layer_norm(raw, (H,), gam, bet, 1e-5) if view else raw
    self_b, g1, b1, bo = 0.3 * rn(H), 1 + 0.2 * rn(H), 0.2 * rn(H), 0.3 * rn(H)
    kf, vf, c = rn(B, S, heads, H, sc=0.08).to(dtype), rn(B, S, heads, H).to(dtype), rn(B, S, heads)
    want_y, knew, vnew, want_attn = _attn_reference(dtype, heads, t, R, x, qkvf, kc, vfc, anc.long(), self_b, g1, b1, kf, vf, c, bo)

    dev = lambda v: v.to(DEV).contiguous()
    d_qkvf, d_kc, d_vfc, d_anc, d_raw = dev(qkvf), dev(kc), dev(vfc), dev(anc), dev(raw)
    d_stats, d_gam, d_bet = dev(_stats(raw)), dev(gam), dev(bet)
    d_sb, d_g1, d_b1, d_bo, d_kf, d_vf, d_c = dev(self_b), dev(g1), dev(b1), dev(bo), dev(kf), dev(vf), dev(c)
    y = torch.full((N, H), float("nan"), device=DEV)
    st_out = torch.full((N, H // 128, 2), float("nan"), device=DEV)
    attn = torch.full((N, S), 7.0, device=DEV)
    d = L.AttnFusedDesc()
    d.qkvf, d.kcache, d.vfcache, d.anc = d_qkvf.data_ptr(), d_kc.data_ptr(), d_vfc.data_ptr(), d_anc.data_ptr()
    if view:
        d.x_ln.raw, d.x_ln.stats, d.x_ln.gamma, d.x_ln.beta = d_raw.data_ptr(), d_stats.data_ptr(), d_gam.data_ptr(), d_bet.data_ptr()
        d.x_ln.ld, d.x_ln.parts, d.x_ln.width = H, H // 128, H
    else:
        d.x = d_raw.data_ptr()
    d.self_b, d.ln1_g, d.ln1_b, d.kf, d.vf, d.c, d.bo = (v.data_ptr() for v in (d_sb, d_g1, d_b1, d_kf, d_vf, d_c, d_bo))
    d.y, d.stats_out, d.attn_out, d.ld_attn, d.accumulate, d.attn_scale = y.data_ptr(), st_out.data_ptr(), attn.data_ptr(), S, 0, 1.0
    d.dtype, d.N, d.H, d.heads, d.S, d.rows_per_image, d.t, d.t_cap = L.dtype_code(dtype), N, H, heads, S, R, t, t_cap
    assert L.lib.mic_attn_fused_supported(d.dtype, H, heads, S, R, t_cap) == 1
    L.check(L.lib.mic_attn_fused(C.byref(d), L.stream_ptr()), "mic_attn_fused")
    torch.cuda.synchronize()
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)  # bf16 operands, fp32 arithmetic
    torch.testing.assert_close(y.cpu().double(), want_y, **tol)
    torch.testing.assert_close(attn.cpu().double(), want_attn, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(st_out.cpu(), _stats(y.cpu()), rtol=1e-3, atol=1e-2)
    # position t of both caches holds the fresh projections, rounded; every other position is untouched
    torch.testing.assert_close(d_kc[:, t].cpu().double(), knew, rtol=0, atol=0)
    torch.testing.assert_close(d_vfc[:, t].cpu().double(), torch.stack(vnew, 1), rtol=0, atol=0)
    keep = [p for p in range(t_cap) if p != t]
    bits = lambda v: v.contiguous().view(torch.int16 if dtype == torch.bfloat16 else torch.int32)  # NaN cells compare by bit pattern
    assert torch.equal(bits(d_kc[:, keep].cpu()), bits(kc[:, keep])) and torch.equal(bits(d_vfc[:, keep].cpu()), bits(vfc[:, keep]))
    # in place (y aliasing the raw stream, as the step driver calls it) and accumulating attention
    if view:
        d.y, d.accumulate, d.attn_scale = d_raw.data_ptr(), 1, 0.5
        L.check(L.lib.mic_attn_fused(C.byref(d), L.stream_ptr()), "mic_attn_fused")
        torch.cuda.synchronize()
        assert torch.equal(d_raw, y)
        torch.testing.assert_close(attn.cpu().double(), 1.5 * want_attn, rtol=1e-3, atol=1e-5)


def test_attn_fused_rejects_bad_arguments(amd):
    L = amd._lib
    assert L.lib.mic_attn_fused_supported(L.BF16, 512, 2, 49, 3, 31) == 1
    assert L.lib.mic_attn_fused_supported(L.BF16, 512, 2, 49, 3, 41) == 1   # the reference's default max_caption_len
    assert L.lib.mic_attn_fused_supported(L.BF16, 512, 2, 49, 3, 49) == 0   # more than 48 cached positions
    assert L.lib.mic_attn_fused_supported(L.BF16, 256, 2, 49, 3, 31) == 0   # rows of 512 bytes
    assert L.lib.mic_attn_fused_supported(L.F32, 256, 4, 49, 3, 31) == 0    # more than two heads
    assert L.lib.mic_attn_fused_supported(L.BF16, 512, 2, 65, 3, 31) == 0
    d = L.AttnFusedDesc()
    assert L.lib.mic_attn_fused(C.byref(d), None) == -1 and b"null pointer" in L.lib.mic_last_error()


def _beam_state(L, N, t_cap, g, t, eos, finished_rows=()):
    score = (torch.randn(N, generator=g, dtype=torch.float64) - 3.0)
    ln = torch.full((N,), t + 1, dtype=torch.int32)
    seq = torch.randint(4, 50, (N, t_cap), generator=g).int()
    anc = torch.randint(0, 1, (N, t_cap), generator=g).int()
    for r in finished_rows:
        ln[r] = max(2, t)
        seq[r, ln[r] - 1] = eos
    return score, ln, seq, anc


@pytest.mark.parametrize("K,B,V,t,dtype", [(3, 6, 3500, 4, torch.bfloat16), (5, 4, 3500, 9, torch.bfloat16), (1, 9, 200, 2, torch.float32),
                                            (8, 3, 4096, 6, torch.float32), (3, 5, 777, 0, torch.bfloat16), (4, 3, 1000, 3, torch.float32)])
def test_beam_tail_equals_the_three_launches(amd, K, B, V, t, dtype):
    """Same logits, same state: the one-launch tail must leave identical beam state (scores bit for bit up to four beams),
    parents and live count, and the embedding rows of live beams must equal mic_embed_pos on the tokens it chose."""
    L = amd._lib
    N, t_cap, H, eos = B * K, 16, 64, 2
    g = torch.Generator().manual_seed(50 + K + t)
    logits = (torch.randn(N, V, generator=g) * 3).to(DEV)
    logits[:, eos] += 4.0  # some beams end
    if V > 600:
        logits[1, 500] = logits[1, 77] = logits[1].max() + 1  # an exact tie inside a row: the lower index ranks first
    fin = [r for r in (1, K) if r < N and t >= 2]
    score, ln, seq, anc = _beam_state(L, N, t_cap, g, t, eos, fin)
    tok_emb, pos_emb = torch.randn(V, H, generator=g).to(DEV), torch.randn(t_cap, H, generator=g).to(DEV)
    sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(DEV)
    ns = 1 if t == 0 else K

    def run(fused):
        cur, nxt = L.BeamState(), L.BeamState()
        bufs = [tuple(v.clone().to(DEV) for v in (score, ln, seq, anc)), tuple(torch.zeros_like(v).to(DEV) for v in (score, ln, seq, anc))]
        for st, b in zip((cur, nxt), bufs):
            st.score, st.len, st.seq, st.anc = (v.data_ptr() for v in b)
        parent = torch.full((N,), -1, dtype=torch.int32, device=DEV)
        live = torch.zeros(1, dtype=torch.int32, device=DEV)
        x = torch.zeros(N, H, device=DEV)
        xt = torch.zeros(N, H, device=DEV, dtype=dtype)
        if fused:
            L.check(L.lib.mic_beam_tail(logits.data_ptr(), V, V, C.byref(cur), C.byref(nxt), parent.data_ptr(), live.data_ptr(),
                                        sqrt_tab.data_ptr(), B, K, ns, t, t_cap, eos, tok_emb.data_ptr(), pos_emb.data_ptr(), x.data_ptr(),
                                        xt.data_ptr(), L.dtype_code(dtype), H, V, L.stream_ptr()), "mic_beam_tail")
        else:
            tv, ti = torch.empty(N, K, device=DEV), torch.empty(N, K, dtype=torch.int32, device=DEV)
            L.check(L.lib.mic_logsoftmax_topk(logits.data_ptr(), V, N, V, K, 1.0, 0, tv.data_ptr(), ti.data_ptr(), None, L.stream_ptr()))
            L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tv.data_ptr(), ti.data_ptr(), parent.data_ptr(), live.data_ptr(),
                                          sqrt_tab.data_ptr(), B, K, ns, t, t_cap, eos, 0, L.stream_ptr()))
            L.check(L.lib.mic_embed_pos(bufs[1][2].data_ptr(), t_cap, t + 1, tok_emb.data_ptr(), pos_emb.data_ptr(), x.data_ptr(),
                                        xt.data_ptr(), H, L.dtype_code(dtype), N, H, V, L.stream_ptr()))
        torch.cuda.synchronize()
        return [v.cpu() for v in bufs[1]] + [parent.cpu(), live.cpu(), x.cpu(), xt.cpu().float()]

    a, b = run(True), run(False)
    for i in range(6):
        if i == 0 and K > 4:
            # beyond four beams a row is summed by 128 threads instead of 256: the log-probabilities agree to fp32 rounding
            torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-6)
            continue
        assert torch.equal(a[i], b[i]), f"state field {i} differs"
    live_rows = a[1] == t + 2  # beams that were extended: their new token sits at position t + 1
    assert live_rows.any()
    assert torch.equal(a[6][live_rows], b[6][live_rows]) and torch.equal(a[7][live_rows], b[7][live_rows])
    assert torch.isfinite(a[6]).all()


def _model(amd, dtype, H, heads, V=300, Ly=2, PT=40, seed=3):
    sd = amd.weights.random_transformer_state(V, H, Ly, PT, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in list(sd):
        if k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("layer_norm.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    return sd


class _Vocab:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def str_to_idx(self, s):
        return {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2, "<UNK>": 3}[s]

    def encode_as_words(self, ids):
        return [int(i) for i in ids]


@pytest.mark.parametrize("beam,max_len", [(3, 30), (5, 12), (1, 20), (3, 40)])
def test_fused_step_fp32_ids_equal_oracle(amd, beam, max_len):
    """fp32 at H=256 (the hidden size whose rows the fused attention kernel takes in fp32), 2 heads: ids identical to the
    fp32 oracle, log-probabilities within 1e-3, on the schedule the benchmark times (the one-launch step for up to three
    beams, one launch per sub-layer above); max_len 40, the reference's default, takes it too (41 cached positions)."""
    from oracle import captioner_oracle as O
    V, H, heads, B, S = 300, 256, 2, 5, 49
    sd = _model(amd, torch.float32, H, heads, V)
    m = amd.transformer.ImageCaptioner(_Vocab(V), heads, torch.float32, encoder="none", sync_every=max_len).load_state_dict(sd).to(DEV)
    feats_bchw = torch.randn(B, H, 7, 7, generator=torch.Generator().manual_seed(8)).relu()
    feats = O.tokens_major(feats_bchw)
    eff = min(max_len, 40)
    t_cap = eff + 1 if beam > 1 else eff
    ws, t = m._workspace(B * beam, t_cap, S, beam, beam, 0)
    assert "qkvf" in t and m.step_schedule(ws) == ("persistent" if beam <= 3 else "fused14")
    ids, lp = m.generate_from_features(feats.to(DEV), _Vocab(V), max_len, None, beam)[:2]
    assert m.last_schedule == ("persistent" if beam <= 3 else "fused14")
    want_ids, want_lp = O.transformer_generate(sd, heads, feats_bchw, max_len, None, beam)[:2]
    assert ids == [list(map(int, w)) for w in want_ids]
    torch.testing.assert_close(lp.cpu(), torch.as_tensor(want_lp, dtype=torch.float32), rtol=1e-3, atol=1e-3)


def test_fused_step_bf16_tracks_unfused(amd, monkeypatch):
    """bf16 at the benchmark's decoder size: the fused-attention step against the unfused chain on the same features:
    teacher-forced logits within bf16 rounding of each other, and beam-search attention output present."""
    V, H, heads, B, S, T = 500, 512, 2, 6, 49, 12
    sd = _model(amd, torch.bfloat16, H, heads, V, Ly=3)
    m = amd.transformer.ImageCaptioner(_Vocab(V), heads, torch.bfloat16, encoder="none", sync_every=T).load_state_dict(sd).to(DEV)
    feats = torch.randn(B, S, H, generator=torch.Generator().manual_seed(18)).to(DEV)
    caps = torch.randint(4, V, (B, T), generator=torch.Generator().manual_seed(19))
    fused = m.forward_from_features(feats, caps).cpu()
    ws, t = m._workspace(B, T, S, 1, 1, 0)
    assert "qkvf" in t
    saved = ws.qkvf
    ws.qkvf = None  # the same workspace without the packed-projection buffer takes the unfused chain
    plain = m.forward_from_features(feats, caps).cpu()
    ws.qkvf = saved
    from oracle import captioner_oracle as O
    want = O.transformer_decode(sd, heads, feats.float().cpu(), caps.long())
    want = want[0] if isinstance(want, tuple) else want
    err = (fused - plain).abs().max().item()
    scale = plain.abs().max().item()
    assert err < 0.03 * scale, (err, scale)
    if want is not None:
        e_f, e_p = (fused - want).abs().mean().item(), (plain - want).abs().mean().item()
        assert e_f < 2.0 * e_p + 1e-3, (e_f, e_p)
    toks, lens, score, attn = m.beam_search_ids(feats, T, 3, return_attn=True)
    assert attn is not None and torch.isfinite(attn).all() and torch.isfinite(score).all()
    a = attn.cpu()
    for b in range(B):
        for p in range(int(lens[b]) - 1):
            assert abs(float(a[b, p].sum()) - 1.0) < 1e-3


@pytest.mark.parametrize("K,B,V,t,dtype", [(3, 6, 3500, 4, torch.bfloat16), (5, 4, 1000, 7, torch.float32), (1, 5, 300, 3, torch.bfloat16)])
def test_beam_tail_in_place_with_state_gathers(amd, K, B, V, t, dtype):
    """The general form of the tail as the LSTM search uses it: finished beams stay in their slots, the embedding has no
    positional part and lands in the input half of a wider [x | h] matrix, and four per-beam state buffers (two with a row
    stride larger than the bytes moved) follow the survivors.  Reference: top-k + in-place selection + mic_gather_rows per
    buffer + mic_embed_pos."""
    L = amd._lib
    N, t_cap, E, Hd, eos = B * K, 16, 32, 48, 2
    kpad = E + Hd + 16
    g = torch.Generator().manual_seed(80 + K + t)
    logits = (torch.randn(N, V, generator=g) * 3).to(DEV)
    logits[:, eos] += 4.0
    fin = [r for r in (1, K) if r < N]
    score, ln, seq, anc = _beam_state(L, N, t_cap, g, t, eos, fin)
    tok_emb = torch.randn(V, E, generator=g).to(DEV)
    sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(DEV)
    c_src = [torch.randn(N, Hd, generator=g).to(DEV) for _ in range(2)]
    xh_src = [torch.randn(N, kpad, generator=g).to(DEV, dtype) for _ in range(2)]
    sz = xh_src[0].element_size()

    def run(fused):
        cur, nxt = L.BeamState(), L.BeamState()
        bufs = [tuple(v.clone().to(DEV) for v in (score, ln, seq)), tuple(torch.zeros_like(v).to(DEV) for v in (score, ln, seq))]
        for st, b in zip((cur, nxt), bufs):
            st.score, st.len, st.seq = (v.data_ptr() for v in b)
        parent = torch.full((N,), -1, dtype=torch.int32, device=DEV)
        live = torch.zeros(1, dtype=torch.int32, device=DEV)
        c_dst = [torch.zeros_like(v) for v in c_src]
        xh_dst = [torch.zeros_like(v) for v in xh_src]
        if fused:
            d = L.BeamTailDesc()
            d.logits, d.ld, d.V, d.cur, d.nxt = logits.data_ptr(), V, V, C.addressof(cur), C.addressof(nxt)
            d.parent, d.live_images, d.sqrt_tab = parent.data_ptr(), live.data_ptr(), sqrt_tab.data_ptr()
            d.B, d.K, d.n_live_slots, d.t, d.t_cap, d.eos_id, d.finished_in_place = B, K, K, t, t_cap, eos, 1
            d.tok_emb, d.xt, d.ld_xt, d.dtype, d.H, d.vocab = tok_emb.data_ptr(), xh_dst[0].data_ptr(), kpad, L.dtype_code(dtype), E, V
            for l in range(2):
                d.gather_src[2 * l], d.gather_dst[2 * l] = c_src[l].data_ptr(), c_dst[l].data_ptr()
                d.gather_row_bytes[2 * l] = d.gather_stride_bytes[2 * l] = Hd * 4
                d.gather_src[2 * l + 1], d.gather_dst[2 * l + 1] = xh_src[l].data_ptr() + E * sz, xh_dst[l].data_ptr() + E * sz
                d.gather_row_bytes[2 * l + 1], d.gather_stride_bytes[2 * l + 1] = Hd * sz, kpad * sz
            d.n_gather = 4
            L.check(L.lib.mic_beam_tail_ex(C.byref(d), L.stream_ptr()), "mic_beam_tail_ex")
        else:
            tv, ti = torch.empty(N, K, device=DEV), torch.empty(N, K, dtype=torch.int32, device=DEV)
            L.check(L.lib.mic_logsoftmax_topk(logits.data_ptr(), V, N, V, K, 1.0, 0, tv.data_ptr(), ti.data_ptr(), None, L.stream_ptr()))
            L.check(L.lib.mic_beam_select(C.byref(cur), C.byref(nxt), tv.data_ptr(), ti.data_ptr(), parent.data_ptr(), live.data_ptr(),
                                          sqrt_tab.data_ptr(), B, K, K, t, t_cap, eos, 1, L.stream_ptr()))
            full = [torch.zeros_like(v) for v in xh_src]
            srcs = (C.c_void_p * 4)(c_src[0].data_ptr(), xh_src[0].data_ptr(), c_src[1].data_ptr(), xh_src[1].data_ptr())
            dsts = (C.c_void_p * 4)(c_dst[0].data_ptr(), full[0].data_ptr(), c_dst[1].data_ptr(), full[1].data_ptr())
            nbytes = (C.c_int32 * 4)(Hd * 4, kpad * sz, Hd * 4, kpad * sz)
            L.check(L.lib.mic_gather_rows_multi(srcs, dsts, nbytes, 4, parent.data_ptr(), N, L.stream_ptr()), "mic_gather_rows_multi")
            for l in range(2):
                xh_dst[l][:, E:E + Hd] = full[l][:, E:E + Hd]
            L.check(L.lib.mic_embed_pos(bufs[1][2].data_ptr(), t_cap, t + 1, tok_emb.data_ptr(), None, None,
                                        xh_dst[0].data_ptr(), kpad, L.dtype_code(dtype), N, E, V, L.stream_ptr()))
        torch.cuda.synchronize()
        return [v.cpu() for v in bufs[1]] + [parent.cpu(), live.cpu()] + [v.cpu() for v in c_dst] + [v.float().cpu() for v in xh_dst]

    a, b = run(True), run(False)
    for i in range(5):
        if i == 0 and K > 4:
            torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-6)
            continue
        assert torch.equal(a[i], b[i]), f"state field {i} differs"
    par = a[3].long()
    assert torch.equal(a[5], c_src[0].cpu()[par]) and torch.equal(a[6], c_src[1].cpu()[par]) and torch.equal(a[5], b[5])
    assert torch.equal(a[8], b[8])                        # layer 1: recurrent half moved, the rest untouched (zeros)
    grew = a[1] == t + 2                                  # beams that were extended got an embedding row
    assert grew.any() and torch.equal(a[7][grew], b[7][grew])
    assert torch.equal(a[7][:, E:], b[7][:, E:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Hd", [(7, 512), (64, 256), (5, 96), (3, 1024)])
def test_lstm_cell_with_layernorm_equals_the_two_launches(amd, N, Hd, dtype):
    L = amd._lib
    g = torch.Generator().manual_seed(N + Hd)
    gates = torch.randn(N, 4 * Hd, generator=g).to(DEV)
    c0 = torch.randn(N, Hd, generator=g).to(DEV)
    ga, be = (1 + 0.3 * torch.randn(Hd, generator=g)).to(DEV), (0.3 * torch.randn(Hd, generator=g)).to(DEV)
    ld0, ldn = Hd + 8, Hd + 16
    assert L.lib.mic_lstm_cell_ln_supported(Hd) == 1 and L.lib.mic_lstm_cell_ln_supported(1030) == 0
    out = []
    for fused in (True, False):
        c, h = c0.clone(), torch.zeros(N, Hd, device=DEV)
        ht0 = torch.zeros(N, ld0, device=DEV, dtype=dtype)
        hn, hnt = torch.zeros(N, Hd, device=DEV), torch.zeros(N, ldn, device=DEV, dtype=dtype)
        if fused:
            L.check(L.lib.mic_lstm_cell_ln(gates.data_ptr(), c.data_ptr(), h.data_ptr(), ht0.data_ptr(), ld0, ga.data_ptr(), be.data_ptr(),
                                           hn.data_ptr(), hnt.data_ptr(), ldn, L.dtype_code(dtype), N, Hd, L.stream_ptr()), "mic_lstm_cell_ln")
        else:
            L.check(L.lib.mic_lstm_cell(gates.data_ptr(), c.data_ptr(), h.data_ptr(), ht0.data_ptr(), ld0, None, 0, L.dtype_code(dtype), N, Hd,
                                        L.stream_ptr()))
            L.check(L.lib.mic_layernorm(h.data_ptr(), ga.data_ptr(), be.data_ptr(), hn.data_ptr(), hnt.data_ptr(), ldn, L.dtype_code(dtype), N, Hd,
                                        L.stream_ptr()))
        torch.cuda.synchronize()
        out.append([v.float().cpu() for v in (c, h, ht0, hn, hnt)])
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    for x, y in zip(*out):
        torch.testing.assert_close(x, y, **tol)
    i, f, gg, o = gates.cpu().double().chunk(4, dim=1)
    cn = torch.sigmoid(f) * c0.cpu().double() + torch.sigmoid(i) * torch.tanh(gg)
    hn_ref = torch.nn.functional.layer_norm(torch.sigmoid(o) * torch.tanh(cn), (Hd,), ga.cpu().double(), be.cpu().double(), 1e-5)
    torch.testing.assert_close(out[0][3].double(), hn_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,E,Hd", [(192, 512, 512), (37, 256, 128), (5, 64, 96)])
def test_lstm_gate_gemm_with_the_cell_in_its_epilogue(amd, N, E, Hd, dtype):
    """mic_lstm_gates_cell (unit-major weight rows, the cell in the GEMM's epilogue) against the gate GEMM in nn.LSTM's row order
    followed by mic_lstm_cell: the same accumulators meet the same cell arithmetic (its one multiply-add spelt out as an fma
    in both kernels), so c', h' and the rounded copies are equal bit for bit, and they match the cell in float64."""
    L, W = amd._lib, amd.weights
    g = torch.Generator().manual_seed(N + Hd)
    km = W.k_multiple(dtype)
    K = (E + Hd + km - 1) // km * km
    w = torch.randn(4 * Hd, E + Hd, generator=g) / (E + Hd) ** 0.5
    b = 0.1 * torch.randn(4 * Hd, generator=g)
    xh = torch.zeros(N, K, dtype=dtype)
    xh[:, :E + Hd] = torch.randn(N, E + Hd, generator=g).to(dtype)
    c0 = torch.randn(N, Hd, generator=g)
    perm = ((torch.arange(4 * Hd).view(Hd, 4) % 4) * Hd + torch.arange(Hd).view(Hd, 1)).reshape(-1)
    wpad = torch.nn.functional.pad(w, (0, K - E - Hd))
    wf = W.frag_matrix(W._pad_rows(wpad), dtype, DEV)
    wf_cell = W.frag_matrix(W._pad_rows(wpad[perm]), dtype, DEV)
    d_b, d_bc, d_xh = b.to(DEV), b[perm].contiguous().to(DEV), xh.to(DEV)
    ld0, ld1 = Hd + 8, Hd + 24
    out = []
    for fused in (True, False):
        c, h = c0.clone().to(DEV), torch.zeros(N, Hd, device=DEV)
        t0, t1 = torch.zeros(N, ld0, device=DEV, dtype=dtype), torch.zeros(N, ld1, device=DEV, dtype=dtype)
        if fused:
            d = L.SkinnyDesc()
            d.x, d.ld_x, d.wf, d.bias = d_xh.data_ptr(), K, wf_cell.data_ptr(), d_bc.data_ptr()
            d.dtype, d.out_dtype, d.M, d.N, d.K = L.dtype_code(dtype), L.F32, N, 4 * Hd, K
            L.check(L.lib.mic_lstm_gates_cell(C.byref(d), c.data_ptr(), h.data_ptr(), t0.data_ptr(), ld0, t1.data_ptr(), ld1, Hd, L.stream_ptr()),
                    "mic_lstm_gates_cell")
        else:
            gates = torch.empty(N, 4 * Hd, device=DEV)
            L.check(L.lib.mic_linear_skinny(d_xh.data_ptr(), K, wf.data_ptr(), d_b.data_ptr(), gates.data_ptr(), L.F32, 4 * Hd, L.dtype_code(dtype),
                                            N, 4 * Hd, K, 0, L.stream_ptr()), "mic_linear_skinny")
            L.check(L.lib.mic_lstm_cell(gates.data_ptr(), c.data_ptr(), h.data_ptr(), t0.data_ptr(), ld0, t1.data_ptr(), ld1, L.dtype_code(dtype), N, Hd,
                                        L.stream_ptr()), "mic_lstm_cell")
        torch.cuda.synchronize()
        out.append([v.float().cpu() for v in (c, h, t0, t1)])
    for name, x, y in zip(("c", "h", "ht0", "ht1"), *out):  # the cell is spelt the same way in both kernels: not a bit differs
        assert torch.equal(x, y), f"{name}: max |d| = {(x - y).abs().max().item()}"
    assert torch.equal(out[0][2][:, Hd:], torch.zeros(N, ld0 - Hd)) and torch.equal(out[0][3][:, Hd:], torch.zeros(N, ld1 - Hd))
    pre = xh.double() @ wpad.to(dtype).double().t() + b.double()
    i_, f_, g_, o_ = pre.chunk(4, dim=1)
    cn = torch.sigmoid(f_) * c0.double() + torch.sigmoid(i_) * torch.tanh(g_)
    hn = torch.sigmoid(o_) * torch.tanh(cn)
    t32 = dict(rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(out[0][0].double(), cn, **t32)
    torch.testing.assert_close(out[0][1].double(), hn, **t32)
    # h' must not land in the rows the launch is reading
    d = L.SkinnyDesc()
    d.x, d.ld_x, d.wf, d.bias = d_xh.data_ptr(), K, wf_cell.data_ptr(), d_bc.data_ptr()
    d.dtype, d.out_dtype, d.M, d.N, d.K = L.dtype_code(dtype), L.F32, N, 4 * Hd, K
    c, h = c0.clone().to(DEV), torch.zeros(N, Hd, device=DEV)
    rc = L.lib.mic_lstm_gates_cell(C.byref(d), c.data_ptr(), h.data_ptr(), d_xh.data_ptr() + E * d_xh.element_size(), K, None, 0, Hd, L.stream_ptr())
    assert rc != 0 and b"overwrite" in L.lib.mic_last_error()
