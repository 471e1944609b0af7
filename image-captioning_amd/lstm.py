"""LSTM captioners (basic, intermediate) on libmic_hip: host-side mirror of
models/image_captioner.py:12-159 with the decoders of models/basic.py:8-50 and
models/intermediate.py:10-67.

`generate` keeps the reference's signature and returns ``(captions, None)``
(models/image_captioner.py:61-69).  The recurrent state (h, c) of every beam is carried on the
device and re-ordered by parent index after each selection step instead of re-running the prefix.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from . import weights as W
from .encoders import ResNet50Encoder
from .vocabulary import EOS, PAD, SOS


class ImageCaptioner:
    kind = None  # "basic" | "intermediate"

    def __init__(self, vocab=None, dtype: torch.dtype = torch.bfloat16, encoder: str = "resnet50", sync_every: int = 8):
        if self.kind is None:
            raise TypeError("use BasicImageCaptioner or IntermediateImageCaptioner")
        if encoder not in ("resnet50", "none"):
            raise ValueError(f"Encoder {encoder} not recognized")
        self.vocab, self.dtype, self.encoder_kind = vocab, dtype, encoder
        self.sync_every = max(1, int(sync_every))
        self._sd, self.device, self.encoder, self.dec = None, None, None, None

    def load_state_dict(self, state_dict, strict: bool = True):
        need = ["decoder.embed.weight", "decoder.lstm.weight_ih_l0", "decoder.linear.weight"]
        if self.kind == "intermediate":
            need += ["decoder.init_h.weight", "decoder.init_c.weight", "decoder.layer_norm.weight"]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"state dict is missing {self.kind} decoder parameters: {missing}")
        if self.encoder_kind == "resnet50" and "encoder.resnet.0.weight" not in state_dict:
            raise KeyError("state dict has no ResNet-50 encoder (encoder.resnet.*)")
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        self.device = None
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("image_captioning_amd runs on a HIP device only; there is no CPU path")
        if self._sd is None:
            raise RuntimeError("load_state_dict() must be called before to(device)")
        if self.device == device:
            return self
        with torch.cuda.device(device):
            self.dec, self._keep = W.build_lstm_decoder(self._sd, self.kind, self.dtype, device)
            self.encoder = ResNet50Encoder(self._sd, self.kind, self.dtype, device) if self.encoder_kind == "resnet50" else None
            if self.kind == "intermediate":
                sd = self._sd
                self._init_w = W.matrix(torch.cat([sd["decoder.init_h.weight"], sd["decoder.init_c.weight"]], 0), self.dtype, device)
                self._init_b = W.vector(torch.cat([sd["decoder.init_h.bias"], sd["decoder.init_c.bias"]], 0), device)
        self.device = device
        return self

    def eval(self):
        return self

    @classmethod
    def from_state_dict(cls, state_dict, vocab=None, dtype=torch.bfloat16, device="cuda", encoder: Optional[str] = None, **kw):
        if encoder is None:
            encoder = "resnet50" if "encoder.resnet.0.weight" in state_dict else "none"
        return cls(vocab, dtype, encoder, **kw).load_state_dict(state_dict).to(device)

    # ---- workspace ----------------------------------------------------------------------------
    def _new_ws(self, N: int, shared=None):
        dec, d, dev = self.dec, self.dtype, self.device
        f32 = torch.float32
        t = {"xh": [torch.zeros(N, dec.kpad[l], dtype=d, device=dev) for l in range(dec.layers)],
             "h": [torch.zeros(N, dec.Hd, dtype=f32, device=dev) for _ in range(dec.layers)],
             "c": [torch.zeros(N, dec.Hd, dtype=f32, device=dev) for _ in range(dec.layers)]}
        if shared is None:
            shared = {"gates": torch.empty(N, 4 * dec.Hd, dtype=f32, device=dev), "hn": torch.empty(N, dec.Hd, dtype=f32, device=dev),
                      "hnt": torch.zeros(N, dec.out_k, dtype=d, device=dev), "logits": torch.empty(N, dec.vocab, dtype=f32, device=dev)}
        t.update(shared)
        ws = L.LstmWorkspace()
        ws.N = N
        for l in range(dec.layers):
            ws.xh[l], ws.h[l], ws.c[l] = t["xh"][l].data_ptr(), t["h"][l].data_ptr(), t["c"][l].data_ptr()
        ws.gates, ws.hn, ws.hnt, ws.logits = (t[k].data_ptr() for k in ("gates", "hn", "hnt", "logits"))
        return ws, t, shared

    def _init_state(self, feats: torch.Tensor, rep: int, ws, t):
        """h0/c0: zeros (basic) or init_h/init_c(feature) split per layer (models/intermediate.py:55-56)."""
        dec = self.dec
        N = feats.shape[0] * rep
        for l in range(dec.layers):
            t["h"][l].zero_(); t["c"][l].zero_(); t["xh"][l].zero_()
        stream = L.stream_ptr()
        frep = feats.repeat_interleave(rep, dim=0).contiguous() if rep > 1 else feats
        if self.kind == "intermediate":
            LH = dec.layers * dec.Hd
            hc = torch.empty(N, 2 * LH, dtype=torch.float32, device=self.device)
            d = L.ConvDesc()
            d.in_, d.w, d.bias, d.out = frep.data_ptr(), self._init_w.data_ptr(), self._init_b.data_ptr(), hc.data_ptr()
            d.dtype, d.out_dtype = dec.dtype, L.F32
            d.M, d.N, d.K = N, 2 * LH, self._init_w.shape[1]
            d.OH = d.OW = d.IH = d.IW = 1
            d.in_img_stride, d.in_pix_stride, d.kw_taps, d.run, d.stride = frep.shape[1], 4, 1, self._init_w.shape[1], 1
            d.ldc = 2 * LH
            L.check(L.lib.mic_conv2d(C.byref(d), stream), "mic_conv2d(init_h/init_c)")
            for l in range(dec.layers):
                in_dim = dec.E if l == 0 else dec.Hd
                t["h"][l].copy_(hc[:, l * dec.Hd:(l + 1) * dec.Hd])
                t["c"][l].copy_(hc[:, LH + l * dec.Hd: LH + (l + 1) * dec.Hd])
                t["xh"][l][:, in_dim:in_dim + dec.Hd].copy_(t["h"][l])
        else:
            # the image feature is time-step 0 of the LSTM input (models/basic.py:45-48)
            L.check(L.lib.mic_lstm_step(C.byref(self.dec), C.byref(ws), None, 0, 0, frep.data_ptr(), frep.shape[1], stream),
                    "mic_lstm_step(feature)")

    def _check_feats(self, feats):
        if feats.dim() != 2 or feats.shape[1] != self.dec.E:
            raise ValueError(f"expected image features of shape (B,{self.dec.E}), got {tuple(feats.shape)}")
        k = W.k_multiple(self.dtype)
        f = feats.to(self.device, self.dtype)
        if f.shape[1] % k:  # GEMM rows are read in 64-byte tiles
            f = torch.nn.functional.pad(f, (0, k - f.shape[1] % k))
        return f.contiguous()

    # ---- search -------------------------------------------------------------------------------
    def sample_ids(self, feats, max_len: int, temp: Optional[float], uniforms=None):
        """models/image_captioner.py:82-105 -> int32 tokens (B, n_tokens)."""
        feats = self._check_feats(feats)
        B, dev, dec = feats.shape[0], self.device, self.dec
        ws, t, _ = self._new_ws(B)
        self._init_state(feats, 1, ws, t)
        hot = not (temp is None or temp == 0.0)
        t_cap = max_len + 1
        pad, sos, eos = (self.vocab.str_to_idx(x) for x in (PAD, SOS, EOS))
        tokens = torch.full((B, t_cap), pad, dtype=torch.int32, device=dev)
        tokens[:, 0] = sos
        finished = torch.zeros(B, dtype=torch.uint8, device=dev)
        live = torch.zeros(t_cap, dtype=torch.int32, device=dev)
        val = torch.empty(B, dtype=torch.float32, device=dev)
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        u = None
        if hot:
            u = (uniforms if uniforms is not None else torch.rand(max_len, B, device=dev)).to(dev, torch.float32).contiguous()
        stream = L.stream_ptr()
        done, chunk_start = max_len, 0
        for s in range(max_len):
            L.check(L.lib.mic_lstm_step(C.byref(dec), C.byref(ws), tokens.data_ptr(), t_cap, s, None, 0, stream), "mic_lstm_step")
            if hot:
                L.check(L.lib.mic_sample_multinomial(t["logits"].data_ptr(), dec.vocab, B, dec.vocab, 1.0 / temp, u[s].data_ptr(),
                                                     val.data_ptr(), idx.data_ptr(), stream), "mic_sample_multinomial")
            else:
                L.check(L.lib.mic_logsoftmax_topk(t["logits"].data_ptr(), dec.vocab, B, dec.vocab, 1, 1.0, 1, val.data_ptr(),
                                                  idx.data_ptr(), None, stream), "mic_logsoftmax_topk")
            L.check(L.lib.mic_sample_update(None, idx.data_ptr(), tokens.data_ptr(), t_cap, s + 1, finished.data_ptr(), None,
                                            live[s:].data_ptr(), B, pad, eos, stream), "mic_sample_update")
            if (s + 1) % self.sync_every == 0 or s + 1 == max_len:
                lv = live[chunk_start:s + 1].tolist()
                if 0 in lv:  # the reference stops right after the step in which the last row finished
                    done = chunk_start + lv.index(0) + 1
                    break
                chunk_start = s + 1
        return tokens[:, : done + 1]

    def beam_search_ids(self, feats, max_len: int, beam_size: int):
        """models/image_captioner.py:118-159 -> int32 tokens (B, max_len+1) padded with -1, lengths (B,)."""
        if not 1 < beam_size <= L.MAX_BEAM:
            raise ValueError(f"beam_size must be in [2, {L.MAX_BEAM}], got {beam_size}")
        feats = self._check_feats(feats)
        B, dev, dec, K = feats.shape[0], self.device, self.dec, beam_size
        N, t_cap = B * K, max_len + 1
        w0, t0_, shared = self._new_ws(N)
        w1, t1_, _ = self._new_ws(N, shared)
        wss, tts = [w0, w1], [t0_, t1_]
        self._init_state(feats, K, w0, t0_)
        sos, eos = self.vocab.str_to_idx(SOS), self.vocab.str_to_idx(EOS)
        st, keep = (L.BeamState * 2)(), []
        for i in range(2):
            score = torch.zeros(N, dtype=torch.float64, device=dev)
            ln = torch.ones(N, dtype=torch.int32, device=dev)
            seq = torch.full((N, t_cap), sos, dtype=torch.int32, device=dev)
            keep.append((score, ln, seq))
            st[i].score, st[i].len, st[i].seq, st[i].anc = score.data_ptr(), ln.data_ptr(), seq.data_ptr(), None
        parent = torch.empty(N, dtype=torch.int32, device=dev)
        live = torch.zeros(t_cap, dtype=torch.int32, device=dev)
        top_val = torch.empty(N, K, dtype=torch.float32, device=dev)
        top_idx = torch.empty(N, K, dtype=torch.int32, device=dev)
        sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(dev)
        stream = L.stream_ptr()
        sz = torch.empty((), dtype=self.dtype).element_size()
        steps = 0
        for s in range(max_len):
            cur, nxt = s & 1, (s + 1) & 1
            L.check(L.lib.mic_lstm_step(C.byref(dec), C.byref(wss[cur]), keep[cur][2].data_ptr(), t_cap, s, None, 0, stream), "mic_lstm_step")
            L.check(L.lib.mic_logsoftmax_topk(shared["logits"].data_ptr(), dec.vocab, N, dec.vocab, K, 1.0, 0, top_val.data_ptr(),
                                              top_idx.data_ptr(), None, stream), "mic_logsoftmax_topk")
            L.check(L.lib.mic_beam_select(C.byref(st[cur]), C.byref(st[nxt]), top_val.data_ptr(), top_idx.data_ptr(), parent.data_ptr(),
                                          live[s:].data_ptr(), sqrt_tab.data_ptr(), B, K, 1 if s == 0 else K, s, t_cap, eos, 1, stream),
                    "mic_beam_select")
            for l in range(dec.layers):  # survivors inherit their parent's recurrent state
                for name, rb in (("h", dec.Hd * 4), ("c", dec.Hd * 4), ("xh", dec.kpad[l] * sz)):
                    L.check(L.lib.mic_gather_rows(tts[cur][name][l].data_ptr(), tts[nxt][name][l].data_ptr(), parent.data_ptr(), N, rb,
                                                  stream), "mic_gather_rows")
            steps = s + 1
            if steps % self.sync_every == 0 and steps < max_len and int(live[s].item()) == 0:
                break
        tokens = torch.empty(B, t_cap, dtype=torch.int32, device=dev)
        lengths = torch.empty(B, dtype=torch.int32, device=dev)
        score = torch.empty(B, dtype=torch.float32, device=dev)
        L.check(L.lib.mic_beam_finalize(C.byref(st[steps & 1]), sqrt_tab.data_ptr(), B, K, t_cap, tokens.data_ptr(), lengths.data_ptr(),
                                        score.data_ptr(), None, 0, None, stream), "mic_beam_finalize")
        self._keep_alive = (keep, wss, tts)
        return tokens, lengths

    # ---- reference-compatible API -------------------------------------------------------------
    def generate(self, images: torch.Tensor, vocab, max_len: int, temp: Optional[float], beam_size: int, device,
                 no_grad: bool = True, return_attn=None, **kwargs) -> tuple:
        """models/image_captioner.py:40-69: returns (captions, None)."""
        kwargs.pop("return_atnn", None)
        if kwargs:
            raise TypeError(f"generate() got unexpected keyword arguments {sorted(kwargs)}")
        if not no_grad:
            raise NotImplementedError("generate(no_grad=False) is training-only and not supported by the HIP path")
        self.to(device)
        if self.encoder is None:
            raise RuntimeError("this captioner was built without an encoder; pass features to generate_from_features")
        return self.generate_from_features(self.encoder(images), vocab, max_len, temp, beam_size), None

    def generate_from_features(self, feats, vocab, max_len: int, temp, beam_size: int, uniforms=None):
        self.vocab = vocab if vocab is not None else self.vocab
        if beam_size > 1:
            tokens, lengths = self.beam_search_ids(feats, max_len, beam_size)
            tok, ln = tokens.cpu().numpy(), lengths.cpu().numpy()
            return [self.vocab.encode_as_words(tok[b, : ln[b]].tolist()) for b in range(tok.shape[0])]
        tokens = self.sample_ids(feats, max_len, temp, uniforms)
        return [self.vocab.encode_as_words(r) for r in tokens.cpu().numpy().tolist()]


class BasicImageCaptioner(ImageCaptioner):
    """models/basic.py:53-79."""
    kind = "basic"


class IntermediateImageCaptioner(ImageCaptioner):
    """models/intermediate.py:70-96."""
    kind = "intermediate"
