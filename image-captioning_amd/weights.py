"""State-dict -> device weight images for libmic_hip.

Accepts the reference's parameter names (``checkpoint["model_state"]`` written by train.py:390-403,
loaded by plotter/captions.py:393-396) and lays the tensors out the way the kernels read them:
K-contiguous ``[out_features padded to 128][K]`` matrices in the compute dtype, eval BatchNorm
folded into the convolution, fp32 biases / LayerNorm / embedding tables.  Load-time plumbing only.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

SD = Dict[str, torch.Tensor]
RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # width, blocks, stride of first block
ROW_PAD = 128  # weight rows are padded so any tile of the GEMM kernel may load them unguarded


def k_multiple(dtype: torch.dtype) -> int:
    """K granularity of the GEMM kernel: one 64-byte tile row."""
    return 64 // torch.empty((), dtype=dtype).element_size()


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class Keep:
    """Owns the device tensors a ctypes struct points into."""

    def __init__(self):
        self.tensors: List[torch.Tensor] = []

    def __call__(self, t: torch.Tensor) -> int:
        self.tensors.append(t)
        return t.data_ptr()


def matrix(w: torch.Tensor, dtype: torch.dtype, device, k_pad: Optional[int] = None) -> torch.Tensor:
    """[N][K] fp32 -> zero-padded [ceil128(N)][k_pad] in the compute dtype."""
    n, k = w.shape
    kp = k_pad or _round_up(k, k_multiple(dtype))
    out = torch.zeros(_round_up(n, ROW_PAD), kp, dtype=torch.float32)
    out[:n, :k] = w.float()
    return out.to(device=device, dtype=dtype).contiguous()


def frag_matrix(w: torch.Tensor, dtype: torch.dtype, device) -> torch.Tensor:
    """[N][K] -> MFMA fragment order [N/16][K/e][4 chunks][16 rows][16 bytes] (what mic_pack_weight_frag
    writes): each 16-row x 64-byte tile becomes 1 KiB contiguous in lane order."""
    n, k = w.shape
    e = k_multiple(dtype)
    if n % 16 or k % e:
        raise ValueError(f"fragment packing needs N % 16 == 0 and K % {e} == 0, got {n}x{k}")
    t = w.float().to(dtype).view(n // 16, 16, k // e, 4, e // 4).permute(0, 2, 3, 1, 4).contiguous()
    return t.to(device)


def _pad_rows(w: torch.Tensor, mult: int = ROW_PAD) -> torch.Tensor:
    n = _round_up(w.shape[0], mult)
    if n == w.shape[0]:
        return w
    out = torch.zeros(n, w.shape[1], dtype=w.dtype)
    out[: w.shape[0]] = w
    return out


def vector(v: torch.Tensor, device) -> torch.Tensor:
    return v.detach().float().contiguous().to(device)


def fold_bn(conv_w: torch.Tensor, g: torch.Tensor, b: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps: float = 1e-5):
    """conv followed by eval BatchNorm == conv with w*s and bias b - mean*s, s = g/sqrt(var+eps)."""
    s = g.double() / torch.sqrt(var.double() + eps)
    return (conv_w.double() * s[:, None, None, None]).float(), (b.double() - mean.double() * s).float()


def conv_matrix(w: torch.Tensor, dtype, device) -> torch.Tensor:
    """(cout,cin,kh,kw) -> [cout padded][kh][kw][cin]: tap-major, channel-contiguous K."""
    co, ci, kh, kw = w.shape
    return matrix(w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci), dtype, device)


def stem_matrix(w: torch.Tensor, dtype, device) -> torch.Tensor:
    """(64,3,7,7) -> [128][7 rows][8 pixels][4 channels]; pixel 7 and channel 3 are zero."""
    co = w.shape[0]
    m = torch.zeros(co, 7, 8, 4)
    m[:, :, :7, :3] = w.permute(0, 2, 3, 1).float()
    return matrix(m.reshape(co, 224), dtype, device)


def _conv_layer(keep: Keep, w, bias, dtype, device, stride, pad, relu, stem=False) -> L.ConvLayer:
    co, ci, kh, kw = w.shape
    wm = stem_matrix(w, dtype, device) if stem else conv_matrix(w, dtype, device)
    return L.ConvLayer(keep(wm), keep(vector(bias, device)), ci, co, kh, kw, stride, pad, int(relu))


def build_resnet50(sd: SD, prefix: str, dtype: torch.dtype, device, fuse_tails: bool = True) -> tuple:
    """torchvision resnet50 children 0..7 under ``prefix`` ('encoder.resnet.') -> mic_resnet50."""
    keep = Keep()
    net = L.Resnet50()
    net.dtype = L.dtype_code(dtype)

    def cb(conv: str, bn: str):
        return fold_bn(sd[conv + ".weight"], sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"], sd[bn + ".running_var"])

    w, b = cb(prefix + "0", prefix + "1")
    net.stem = _conv_layer(keep, w, b, dtype, device, 2, 3, True, stem=True)
    bi = 0
    for si, (width, blocks, stride) in enumerate(RESNET50_STAGES):
        net.blocks_per_stage[si] = blocks
        for k in range(blocks):
            p = f"{prefix}{4 + si}.{k}."
            s = stride if k == 0 else 1
            w, b = cb(p + "conv1", p + "bn1")
            net.conv[bi][0] = _conv_layer(keep, w, b, dtype, device, 1, 0, True)
            w, b = cb(p + "conv2", p + "bn2")
            net.conv[bi][1] = _conv_layer(keep, w, b, dtype, device, s, 1, True)
            w, b = cb(p + "conv3", p + "bn3")
            net.conv[bi][2] = _conv_layer(keep, w, b, dtype, device, 1, 0, True)  # ReLU after the residual add
            if (p + "downsample.0.weight") in sd:
                w3, b3 = w, b
                w, b = cb(p + "downsample.0", p + "downsample.1")
                net.conv[bi][3] = _conv_layer(keep, w, b, dtype, device, s, 0, False)
                km = k_multiple(dtype)
                if fuse_tails and k == 0 and w3.shape[1] % km == 0 and w.shape[1] % km == 0:
                    # conv3 and the shortcut convolution share one accumulator: [W3 | Wds] over K = width + cin
                    cat = torch.cat([w3.reshape(w3.shape[0], -1), w.reshape(w.shape[0], -1)], dim=1)
                    T_ = net.tail[si]
                    T_.w, T_.bias = keep(matrix(cat, dtype, device)), keep(vector(b3 + b, device))
                    T_.cin_main, T_.cin_skip, T_.cout, T_.skip_stride = w3.shape[1], w.shape[1], w3.shape[0], s
            bi += 1
    return net, keep


def swin_relative_tables(window: int = 8):
    """The two buffers of torchvision's ShiftedWindowAttentionV2: log-spaced relative coordinates
    (1, 2w-1, 2w-1, 2) and the (w^2 * w^2) index of each token pair into that table."""
    r = torch.arange(-(window - 1), window, dtype=torch.float32)
    table = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    table = table / (window - 1) * 8
    table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / 3.0
    c = torch.arange(window)
    coords = torch.stack(torch.meshgrid(c, c, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - 1
    rel[:, :, 1] += window - 1
    rel[:, :, 0] *= 2 * window - 1
    return table, rel.sum(-1).flatten()


def swin_config_of(sd: SD, prefix: str = "encoder.features."):
    """(embed_dim, depths, heads) read off the parameter shapes."""
    embed = sd[prefix + "0.0.weight"].shape[0]
    depths, heads = [], []
    for si in range(4):
        d = 0
        while f"{prefix}{1 + 2 * si}.{d}.attn.qkv.weight" in sd:
            d += 1
        if d == 0:
            raise KeyError(f"state dict has no Swin stage {si} under {prefix}{1 + 2 * si}.*")
        depths.append(d)
        heads.append(int(sd[f"{prefix}{1 + 2 * si}.0.attn.logit_scale"].shape[0]))
    return embed, depths, heads


def build_swin(sd: SD, prefix: str, dtype: torch.dtype, device, window: int = 8) -> tuple:
    """torchvision swin_v2_* ``features`` parameters under ``prefix`` ('encoder.features.') -> mic_swin.
    Input-independent pieces are evaluated here once: the continuous position bias
    16*sigmoid(cpb_mlp(table))[index] per head, exp(min(logit_scale, log 100)), and the zeroed key bias."""
    keep = Keep()
    net = L.Swin()
    embed, depths, heads = swin_config_of(sd, prefix)
    if sum(depths) > L.SWIN_MAX_BLOCKS:
        raise ValueError(f"{sum(depths)} Swin blocks exceed the supported {L.SWIN_MAX_BLOCKS}")
    if embed % 32:
        raise ValueError(f"Swin embed_dim {embed} must be a multiple of 32")
    net.dtype, net.embed_dim = L.dtype_code(dtype), embed
    pw = sd[prefix + "0.0.weight"]
    if tuple(pw.shape[1:]) != (3, 4, 4):
        raise ValueError(f"patch embedding must be a 4x4 convolution over 3 channels, got {tuple(pw.shape)}")
    pm = torch.zeros(embed, 4, 4, 4)
    pm[:, :, :, :3] = pw.permute(0, 2, 3, 1).float()
    net.patch_w = keep(matrix(pm.reshape(embed, 64), dtype, device))
    net.patch_b = keep(vector(sd[prefix + "0.0.bias"], device))
    net.patch_g = keep(vector(sd[prefix + "0.2.weight"], device))
    net.patch_be = keep(vector(sd[prefix + "0.2.bias"], device))
    table0, index0 = swin_relative_tables(window)
    n = window * window
    bi = 0
    for si in range(4):
        C_ = embed << si
        net.depths[si], net.heads[si] = depths[si], heads[si]
        if heads[si] * 32 != C_:
            raise ValueError(f"Swin stage {si}: {heads[si]} heads over {C_} channels is not 32 channels per head")
        for k in range(depths[si]):
            p = f"{prefix}{1 + 2 * si}.{k}."
            B_ = net.block[bi]
            qb = sd[p + "attn.qkv.bias"].detach().float().clone()
            qb[C_:2 * C_] = 0  # the key bias is held at zero (ShiftedWindowAttentionV2)
            B_.qkv_w, B_.qkv_b = keep(matrix(sd[p + "attn.qkv.weight"], dtype, device)), keep(vector(qb, device))
            B_.proj_w, B_.proj_b = keep(matrix(sd[p + "attn.proj.weight"], dtype, device)), keep(vector(sd[p + "attn.proj.bias"], device))
            B_.fc1_w, B_.fc1_b = keep(matrix(sd[p + "mlp.0.weight"], dtype, device)), keep(vector(sd[p + "mlp.0.bias"], device))
            B_.fc2_w, B_.fc2_b = keep(matrix(sd[p + "mlp.3.weight"], dtype, device)), keep(vector(sd[p + "mlp.3.bias"], device))
            for j in (1, 2):
                setattr(B_, f"n{j}_g", keep(vector(sd[p + f"norm{j}.weight"], device)))
                setattr(B_, f"n{j}_b", keep(vector(sd[p + f"norm{j}.bias"], device)))
            table = sd.get(p + "attn.relative_coords_table", table0).float()
            index = sd.get(p + "attn.relative_position_index", index0).long()
            t = torch.relu(table @ sd[p + "attn.cpb_mlp.0.weight"].float().T + sd[p + "attn.cpb_mlp.0.bias"].float())
            t = (t @ sd[p + "attn.cpb_mlp.2.weight"].float().T).view(-1, heads[si])
            bias = 16 * torch.sigmoid(t[index].view(n, n, heads[si]).permute(2, 0, 1))
            B_.bias_tab = keep(vector(bias, device))
            ls = torch.clamp(sd[p + "attn.logit_scale"].detach().float().view(-1), max=math.log(100.0)).exp()
            B_.logit_scale = keep(vector(ls, device))
            bi += 1
        if si < 3:
            p = f"{prefix}{2 + 2 * si}."
            w = sd[p + "reduction.weight"].float()  # columns: [x(0,0) | x(1,0) | x(0,1) | x(1,1)] by (row, col) parity
            w = w.view(2 * C_, 4, C_)[:, [0, 2, 1, 3], :].reshape(2 * C_, 4 * C_)  # -> (kh, kw) raster order
            net.merge[si].w = keep(matrix(w, dtype, device))
            net.merge[si].g = keep(vector(sd[p + "norm.weight"], device))
            net.merge[si].b = keep(vector(sd[p + "norm.bias"], device))
    return net, keep


def build_tr_decoder(sd: SD, heads: int, dtype: torch.dtype, device) -> tuple:
    """models/transformer.py parameter names -> mic_tr_decoder."""
    keep = Keep()
    dec = L.TrDecoder()
    tok = sd["seq_embedding.token_embedding.weight"]
    pos = sd["seq_embedding.pos_embedding.weight"]
    V, H = tok.shape
    layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("decoder_layers."))
    if layers > L.MAX_LAYERS:
        raise ValueError(f"{layers} decoder layers exceed the supported {L.MAX_LAYERS}")
    if H % heads != 0:
        raise ValueError(f"hidden size {H} is not divisible by num_heads {heads}")
    dec.dtype, dec.H, dec.heads, dec.layers, dec.vocab, dec.pos_rows = L.dtype_code(dtype), H, heads, layers, V, pos.shape[0]
    dec.tok_emb, dec.pos_emb = keep(vector(tok, device)), keep(vector(pos, device))
    for i in range(layers):
        p = f"decoder_layers.{i}."
        Y = dec.layer[i]
        w, b = sd[p + "self_attention.mha.in_proj_weight"], sd[p + "self_attention.mha.in_proj_bias"]
        Y.self_in_w, Y.self_in_b = keep(matrix(w, dtype, device)), keep(vector(b, device))
        Y.self_out_w = keep(matrix(sd[p + "self_attention.mha.out_proj.weight"], dtype, device))
        Y.self_out_b = keep(vector(sd[p + "self_attention.mha.out_proj.bias"], device))
        w, b = sd[p + "cross_attention.mha.in_proj_weight"], sd[p + "cross_attention.mha.in_proj_bias"]
        Y.cross_q_w, Y.cross_q_b = keep(matrix(w[:H], dtype, device)), keep(vector(b[:H], device))
        Y.cross_kv_w, Y.cross_kv_b = keep(matrix(w[H:], dtype, device)), keep(vector(b[H:], device))
        Y.cross_out_w = keep(matrix(sd[p + "cross_attention.mha.out_proj.weight"], dtype, device))
        Y.cross_out_b = keep(vector(sd[p + "cross_attention.mha.out_proj.bias"], device))
        Y.ff1_w, Y.ff1_b = keep(matrix(sd[p + "ff.seq.0.weight"], dtype, device)), keep(vector(sd[p + "ff.seq.0.bias"], device))
        Y.ff2_w, Y.ff2_b = keep(matrix(sd[p + "ff.seq.2.weight"], dtype, device)), keep(vector(sd[p + "ff.seq.2.bias"], device))
        e = k_multiple(dtype)
        if L.lib.mic_linear_res_ln_supported(H) and H % e == 0:
            # rows padded to 128 like every fragment-ordered matrix: the skinny GEMM's workgroups span 128
            # columns and fetch the (zero) tiles past H; the fused GEMM+LayerNorm kernel reads the first H rows only
            Y.self_out_wf = keep(frag_matrix(_pad_rows(sd[p + "self_attention.mha.out_proj.weight"]), dtype, device))
            Y.cross_out_wf = keep(frag_matrix(_pad_rows(sd[p + "cross_attention.mha.out_proj.weight"]), dtype, device))
            Y.ff2_wf = keep(frag_matrix(_pad_rows(sd[p + "ff.seq.2.weight"]), dtype, device))
        if H % e == 0:  # fragment-ordered copies (rows zero-padded to 128) for the skinny GEMM
            ipw = sd[p + "self_attention.mha.in_proj_weight"]
            Y.self_in_wf = keep(frag_matrix(_pad_rows(ipw), dtype, device))
            Y.cross_q_wf = keep(frag_matrix(_pad_rows(sd[p + "cross_attention.mha.in_proj_weight"][:H]), dtype, device))
            Y.ff1_wf = keep(frag_matrix(_pad_rows(sd[p + "ff.seq.0.weight"]), dtype, device))
        hd = H // heads
        if heads <= L.FOLD_MAX_HEADS and hd % e == 0:
            # fold the query projection into the cross keys and the output projection into the cross values
            # (mic_cross_attn_folded): per head, K' = k_h (W_q,h / sqrt d) and V' = v_h W_o,h^T are plain GEMMs
            wq = sd[p + "cross_attention.mha.in_proj_weight"][:H].double()
            bq = sd[p + "cross_attention.mha.in_proj_bias"][:H].double()
            wo = sd[p + "cross_attention.mha.out_proj.weight"].double()
            sc = 1.0 / math.sqrt(hd)
            for h in range(heads):
                Y.cross_kf_w[h] = keep(matrix((wq[h * hd:(h + 1) * hd, :].T * sc).float().contiguous(), dtype, device))
                Y.cross_vf_w[h] = keep(matrix(wo[:, h * hd:(h + 1) * hd].float().contiguous(), dtype, device))
            Y.cross_qb_scaled = keep(vector((bq * sc).float(), device))
            # the same fold composed with the key / value projection: K', V' and the per-position constant straight from
            # the image tokens (two GEMMs per layer where projecting k, v and folding them per head took five)
            ipw, ipb = sd[p + "cross_attention.mha.in_proj_weight"].double(), sd[p + "cross_attention.mha.in_proj_bias"].double()
            wk, bk, wv, bv = ipw[H:2 * H], ipb[H:2 * H], ipw[2 * H:], ipb[2 * H:]
            kw_, kb_, vw_, vb_, cu_, c0_ = [], [], [], [], [], []
            for h in range(heads):
                sl = slice(h * hd, (h + 1) * hd)
                kw_.append(sc * wq[sl, :].T @ wk[sl, :])
                kb_.append(sc * wq[sl, :].T @ bk[sl])
                vw_.append(wo[:, sl] @ wv[sl, :])
                vb_.append(wo[:, sl] @ bv[sl])
                cu_.append(sc * wk[sl, :].T @ bq[sl])
                c0_.append(sc * (bk[sl] @ bq[sl]))
            if H % e == 0:
                Y.cross_kfc_w = keep(matrix(torch.cat(kw_, dim=0).float().contiguous(), dtype, device))
                Y.cross_vfc_w = keep(matrix(torch.cat(vw_, dim=0).float().contiguous(), dtype, device))
                Y.cross_kfc_b = keep(vector(torch.cat(kb_).float(), device))
                Y.cross_vfc_b = keep(vector(torch.cat(vb_).float(), device))
                Y.cross_cu = keep(vector(torch.cat(cu_).float(), device))
                Y.cross_c0 = keep(vector(torch.stack(c0_).float(), device))
        if heads <= L.FOLD_MAX_HEADS and H % e == 0:
            # fold the self-attention output projection into the value rows of the packed in-projection
            # (mic_attn_fused): v'_h = (W_o[:, head h] W_v[head h, :]) x, and the value bias moves into the output bias
            wi = sd[p + "self_attention.mha.in_proj_weight"].double()
            bi = sd[p + "self_attention.mha.in_proj_bias"].double()
            wo = sd[p + "self_attention.mha.out_proj.weight"].double()
            bo = sd[p + "self_attention.mha.out_proj.bias"].double()
            rows, ob = [wi[:2 * H]], bo.clone()
            for h in range(heads):
                sl = slice(h * hd, (h + 1) * hd)
                rows.append(wo[:, sl] @ wi[2 * H:][sl, :])
                ob += wo[:, sl] @ bi[2 * H:][sl]
            Y.self_qkvf_wf = keep(frag_matrix(_pad_rows(torch.cat(rows, dim=0).float()), dtype, device))
            Y.self_qkvf_b = keep(vector(torch.cat([bi[:2 * H], torch.zeros(heads * H, dtype=torch.float64)]).float(), device))
            Y.self_of_b = keep(vector(ob.float(), device))
        for j, nm in enumerate(("self_attention", "cross_attention", "ff"), start=1):
            setattr(Y, f"ln{j}_g", keep(vector(sd[p + nm + ".layer_norm.weight"], device)))
            setattr(Y, f"ln{j}_b", keep(vector(sd[p + nm + ".layer_norm.bias"], device)))
    dec.out_w = keep(matrix(sd["output_layer.linear.weight"], dtype, device))
    dec.out_b = keep(vector(sd["output_layer.linear.bias"], device))
    if H % k_multiple(dtype) == 0:
        dec.out_wf = keep(frag_matrix(_pad_rows(sd["output_layer.linear.weight"]), dtype, device))
    return dec, keep


def build_lstm_decoder(sd: SD, kind: str, dtype: torch.dtype, device) -> tuple:
    """models/basic.py / models/intermediate.py decoder parameter names -> mic_lstm_decoder."""
    keep = Keep()
    dec = L.LstmDecoder()
    emb = sd["decoder.embed.weight"]
    V, E = emb.shape
    Hd = sd["decoder.lstm.weight_hh_l0"].shape[1]
    layers = 1 + max(int(k.rsplit("_l", 1)[1]) for k in sd if k.startswith("decoder.lstm.weight_ih_l"))
    if layers > L.MAX_LAYERS:
        raise ValueError(f"{layers} LSTM layers exceed the supported {L.MAX_LAYERS}")
    km = k_multiple(dtype)
    dec.dtype, dec.E, dec.Hd, dec.layers, dec.vocab = L.dtype_code(dtype), E, Hd, layers, V
    dec.kind = {"basic": 0, "intermediate": 1}[kind]
    dec.embed = keep(vector(emb, device))
    for l in range(layers):
        wi, wh = sd[f"decoder.lstm.weight_ih_l{l}"], sd[f"decoder.lstm.weight_hh_l{l}"]
        kp = _round_up(wi.shape[1] + Hd, km)
        dec.kpad[l] = kp
        dec.w[l] = keep(matrix(torch.cat([wi, wh], dim=1), dtype, device, kp))
        # fragment-ordered copy (rows padded to 128, K to kp) for the weight-streaming GEMM of the decode steps
        dec.wf[l] = keep(frag_matrix(_pad_rows(torch.nn.functional.pad(torch.cat([wi, wh], dim=1).float(), (0, kp - wi.shape[1] - Hd))), dtype, device))
        dec.b[l] = keep(vector(sd[f"decoder.lstm.bias_ih_l{l}"] + sd[f"decoder.lstm.bias_hh_l{l}"], device))
        # the same in unit-major row order (row 4u + g = gate g of unit u; nn.LSTM stacks the gates i, f, g, o): the gate GEMM
        # whose epilogue is the cell (mic_lstm_gates_cell)
        perm = (torch.arange(4 * Hd).view(Hd, 4) % 4) * Hd + torch.arange(Hd).view(Hd, 1)
        perm = perm.reshape(-1)
        wcat = torch.nn.functional.pad(torch.cat([wi, wh], dim=1).float(), (0, kp - wi.shape[1] - Hd))
        dec.wf_cell[l] = keep(frag_matrix(_pad_rows(wcat[perm]), dtype, device))
        dec.b_cell[l] = keep(vector((sd[f"decoder.lstm.bias_ih_l{l}"] + sd[f"decoder.lstm.bias_hh_l{l}"])[perm], device))
    if kind == "intermediate":
        dec.ln_g = keep(vector(sd["decoder.layer_norm.weight"], device))
        dec.ln_b = keep(vector(sd["decoder.layer_norm.bias"], device))
    dec.out_k = _round_up(Hd, km)
    dec.out_w = keep(matrix(sd["decoder.linear.weight"], dtype, device, dec.out_k))
    dec.out_wf = keep(frag_matrix(_pad_rows(torch.nn.functional.pad(sd["decoder.linear.weight"].float(), (0, dec.out_k - Hd))), dtype, device))
    dec.out_b = keep(vector(sd["decoder.linear.bias"], device))
    return dec, keep


# --------------------------------------------------------------------------------------------------
# random initialisation under the reference's names (benchmarks and tests; no checkpoint ships)
# --------------------------------------------------------------------------------------------------
def _uniform(g, shape, bound):
    return (torch.rand(*shape, generator=g) * 2 - 1) * bound


def random_resnet50_state(seed: int = 0, prefix: str = "encoder.resnet.") -> SD:
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (co * k * k))

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)

    conv(prefix + "0", 64, 3, 7)
    bn(prefix + "1", 64)
    cin = 64
    for si, (width, blocks, _) in enumerate(RESNET50_STAGES):
        for k in range(blocks):
            p = f"{prefix}{4 + si}.{k}."
            conv(p + "conv1", width, cin, 1); bn(p + "bn1", width)
            conv(p + "conv2", width, width, 3); bn(p + "bn2", width)
            conv(p + "conv3", width * 4, width, 1); bn(p + "bn3", width * 4)
            if k == 0:
                conv(p + "downsample.0", width * 4, cin, 1); bn(p + "downsample.1", width * 4)
            cin = width * 4
    return sd


def random_swin_state(seed: int = 0, embed: int = 96, depths=(2, 2, 18, 2), heads=(3, 6, 12, 24), window: int = 8,
                      prefix: str = "encoder.features.") -> SD:
    """Random Swin-V2 parameters under the names models/encoders/swin.py:29 registers (defaults: the S configuration)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def lin(name, o, i, bias=True, std=0.02):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * std
        if bias:
            sd[name + ".bias"] = 0.02 * torch.randn(o, generator=g)

    def ln(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    sd[prefix + "0.0.weight"] = torch.randn(embed, 3, 4, 4, generator=g) * 0.1
    sd[prefix + "0.0.bias"] = 0.02 * torch.randn(embed, generator=g)
    ln(prefix + "0.2", embed)
    table, index = swin_relative_tables(window)
    for si in range(4):
        C_ = embed * 2 ** si
        for k in range(depths[si]):
            p = f"{prefix}{1 + 2 * si}.{k}."
            lin(p + "attn.qkv", 3 * C_, C_, std=0.05)
            lin(p + "attn.proj", C_, C_, std=0.05)
            sd[p + "attn.logit_scale"] = torch.log(10 * torch.ones(heads[si], 1, 1)) + 0.3 * torch.randn(heads[si], 1, 1, generator=g)
            lin(p + "attn.cpb_mlp.0", 512, 2, std=0.5)
            lin(p + "attn.cpb_mlp.2", heads[si], 512, bias=False, std=0.1)
            sd[p + "attn.relative_coords_table"] = table.clone()
            sd[p + "attn.relative_position_index"] = index.clone()
            ln(p + "norm1", C_)
            ln(p + "norm2", C_)
            lin(p + "mlp.0", 4 * C_, C_, std=0.05)
            lin(p + "mlp.3", C_, 4 * C_, std=0.05)
        if si < 3:
            p = f"{prefix}{2 + 2 * si}."
            lin(p + "reduction", 2 * C_, 4 * C_, bias=False, std=0.05)
            ln(p + "norm", 2 * C_)
    return sd


def output_bias_from_counts(counts: Sequence[float], banned=(0, 1, 3)) -> torch.Tensor:
    """Log-unigram output bias with PAD/SOS/UNK at -1e9 (Output.adapt, models/transformer.py:248-284)."""
    c = torch.tensor([0.0 if i in banned else float(x) for i, x in enumerate(counts)], dtype=torch.float64)
    total = float(c.sum())
    bias = torch.full((len(c),), -1e9, dtype=torch.float32)
    if total == 0:
        bias.zero_()
        bias[list(banned)] = -1e9
        return bias
    pos = c > 0
    bias[pos] = torch.log((c[pos] / total).float())
    return bias


def random_transformer_state(V: int, H: int, layers: int, pos_rows: int, seed: int = 0, counts=None) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd: SD = {"seq_embedding.token_embedding.weight": torch.randn(V, H, generator=g),
              "seq_embedding.pos_embedding.weight": torch.randn(pos_rows, H, generator=g)}
    sd["seq_embedding.token_embedding.weight"][0].zero_()
    for i in range(layers):
        p = f"decoder_layers.{i}."
        for a in ("self_attention", "cross_attention"):
            sd[p + a + ".mha.in_proj_weight"] = _uniform(g, (3 * H, H), math.sqrt(6.0 / (4 * H)))
            sd[p + a + ".mha.in_proj_bias"] = torch.zeros(3 * H)
            sd[p + a + ".mha.out_proj.weight"] = _uniform(g, (H, H), 1 / math.sqrt(H))
            sd[p + a + ".mha.out_proj.bias"] = torch.zeros(H)
        sd[p + "ff.seq.0.weight"] = _uniform(g, (2 * H, H), 1 / math.sqrt(H))
        sd[p + "ff.seq.0.bias"] = _uniform(g, (2 * H,), 1 / math.sqrt(H))
        sd[p + "ff.seq.2.weight"] = _uniform(g, (H, 2 * H), 1 / math.sqrt(2 * H))
        sd[p + "ff.seq.2.bias"] = _uniform(g, (H,), 1 / math.sqrt(2 * H))
        for a in ("self_attention", "cross_attention", "ff"):
            sd[p + a + ".layer_norm.weight"] = torch.ones(H)
            sd[p + a + ".layer_norm.bias"] = torch.zeros(H)
    sd["output_layer.linear.weight"] = _uniform(g, (V, H), 1 / math.sqrt(H))
    if counts is None:
        counts = [0, 50, 50, 0] + [(i % 97) + 1 for i in range(4, V)]
    sd["output_layer.linear.bias"] = output_bias_from_counts(counts)
    return sd


def random_lstm_state(kind: str, V: int, E: int, Hd: int, layers: int, seed: int = 0) -> SD:
    g = torch.Generator().manual_seed(seed)
    b = 1 / math.sqrt(Hd)
    sd: SD = {"decoder.embed.weight": torch.randn(V, E, generator=g)}
    sd["decoder.embed.weight"][0].zero_()
    for l in range(layers):
        ind = E if l == 0 else Hd
        sd[f"decoder.lstm.weight_ih_l{l}"] = _uniform(g, (4 * Hd, ind), b)
        sd[f"decoder.lstm.weight_hh_l{l}"] = _uniform(g, (4 * Hd, Hd), b)
        sd[f"decoder.lstm.bias_ih_l{l}"] = _uniform(g, (4 * Hd,), b)
        sd[f"decoder.lstm.bias_hh_l{l}"] = _uniform(g, (4 * Hd,), b)
    sd["decoder.linear.weight"] = _uniform(g, (V, Hd), b)
    bias = torch.zeros(V)
    bias[[0, 1, 3]] = -1e9
    sd["decoder.linear.bias"] = bias
    if kind == "intermediate":
        for nm in ("init_h", "init_c"):
            sd[f"decoder.{nm}.weight"] = _uniform(g, (layers * Hd, E), 1 / math.sqrt(E))
            sd[f"decoder.{nm}.bias"] = _uniform(g, (layers * Hd,), 1 / math.sqrt(E))
        sd["decoder.layer_norm.weight"] = torch.ones(Hd)
        sd["decoder.layer_norm.bias"] = torch.zeros(Hd)
    return sd


def random_head_state(kind: str, out_dim: int, seed: int = 0, in_dim: int = 2048) -> SD:
    """Encoder head under the reference's names: projection.0 (transformer / swin), linear.0 (intermediate), linear (basic)."""
    g = torch.Generator().manual_seed(seed + 17)
    b = 1 / math.sqrt(in_dim)
    if kind in ("transformer", "swin"):
        return {"encoder.projection.0.weight": _uniform(g, (out_dim, in_dim, 1, 1), b), "encoder.projection.0.bias": _uniform(g, (out_dim,), b)}
    name = "encoder.linear.0" if kind == "intermediate" else "encoder.linear"
    return {name + ".weight": _uniform(g, (out_dim, in_dim), b), name + ".bias": _uniform(g, (out_dim,), b)}
