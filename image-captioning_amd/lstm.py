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

    def __init__(self, vocab=None, dtype: torch.dtype = torch.bfloat16, encoder: str = "resnet50", sync_every: int = 8,
                 use_graph: bool = True):
        if self.kind is None:
            raise TypeError("use BasicImageCaptioner or IntermediateImageCaptioner")
        if encoder not in ("resnet50", "none"):
            raise ValueError(f"Encoder {encoder} not recognized")
        self.vocab, self.dtype, self.encoder_kind = vocab, dtype, encoder
        self.sync_every = max(1, int(sync_every))
        self.use_graph = bool(use_graph)  # replay each run of decode steps as one HIP graph
        self._buffers = {}  # search buffers per shape: stable addresses keep the captured graphs valid
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
        self._buffers = {}
        return self

    def eval(self):
        return self

    def lane_copy(self):
        """A second handle on the same device weights with buffers of its own, for a concurrent lane (lanes.Lanes)."""
        import copy
        if self.device is None:
            raise RuntimeError("lane_copy() needs a model that is on its device")
        m = copy.copy(self)
        m._buffers = {}
        if self.encoder is not None:
            m.encoder = copy.copy(self.encoder)
            m.encoder._ws = None
        return m

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
             "c": [torch.zeros(N, dec.Hd, dtype=f32, device=dev) for _ in range(dec.layers)],
             "hstage": [torch.zeros(N, dec.kpad[l], dtype=d, device=dev) for l in range(dec.layers)]}
        if shared is None:
            shared = {"gates": torch.empty(N, 4 * dec.Hd, dtype=f32, device=dev), "hn": torch.empty(N, dec.Hd, dtype=f32, device=dev),
                      "hnt": torch.zeros(N, dec.out_k, dtype=d, device=dev), "logits": torch.empty(N, dec.vocab, dtype=f32, device=dev)}
        t.update(shared)
        ws = L.LstmWorkspace()
        ws.N = N
        for l in range(dec.layers):
            ws.xh[l], ws.h[l], ws.c[l] = t["xh"][l].data_ptr(), t["h"][l].data_ptr(), t["c"][l].data_ptr()
            ws.hstage[l] = t["hstage"][l].data_ptr()
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

    def _trim_buffers(self, keep: int = 3):
        while len(self._buffers) > keep:  # oldest shape first
            self._buffers.pop(next(iter(self._buffers)))

    # ---- search -------------------------------------------------------------------------------
    def sample_ids(self, feats, max_len: int, temp: Optional[float], uniforms=None):
        """models/image_captioner.py:82-105 -> int32 tokens (B, n_tokens)."""
        feats = self._check_feats(feats)
        B, dev, dec = feats.shape[0], self.device, self.dec
        hot = not (temp is None or temp == 0.0)
        t_cap = max_len + 1
        pad, sos, eos = (self.vocab.str_to_idx(x) for x in (PAD, SOS, EOS))
        key = ("sample", B, t_cap)
        if key not in self._buffers:
            self._trim_buffers()
            ws, t, _ = self._new_ws(B)
            self._buffers[key] = (ws, t, torch.empty(B, t_cap, dtype=torch.int32, device=dev),
                                  torch.empty(B, dtype=torch.uint8, device=dev), torch.empty(t_cap, dtype=torch.int32, device=dev),
                                  torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                                  torch.empty(max_len, B, dtype=torch.float32, device=dev))
        ws, t, tokens, finished, live, val, idx, ubuf = self._buffers[key]
        self._init_state(feats, 1, ws, t)
        tokens.fill_(pad)
        tokens[:, 0] = sos
        finished.zero_()
        live.zero_()
        if hot:
            ubuf.copy_(uniforms[:max_len] if uniforms is not None else torch.rand(max_len, B, device=dev))
        stream = L.stream_ptr()
        done, s = max_len, 0
        while s < max_len:
            e = min(max_len, s + self.sync_every)
            L.check(L.lib.mic_lstm_sample_steps(C.byref(dec), C.byref(ws), tokens.data_ptr(), finished.data_ptr(), live.data_ptr(),
                                                1.0 / temp if hot else 1.0, ubuf.data_ptr() if hot else None, val.data_ptr(),
                                                idx.data_ptr(), s, e, t_cap, pad, eos, int(self.use_graph), stream),
                    "mic_lstm_sample_steps")
            lv = live[s:e].tolist()
            if 0 in lv:  # the reference stops right after the step in which the last row finished
                done = s + lv.index(0) + 1
                break
            s = e
        return tokens[:, : done + 1].clone()

    def beam_search_ids(self, feats, max_len: int, beam_size: int):
        """models/image_captioner.py:118-159 -> int32 tokens (B, max_len+1) padded with -1, lengths (B,)."""
        if not 1 < beam_size <= L.MAX_BEAM:
            raise ValueError(f"beam_size must be in [2, {L.MAX_BEAM}], got {beam_size}")
        feats = self._check_feats(feats)
        B, dev, dec, K = feats.shape[0], self.device, self.dec, beam_size
        N, t_cap = B * K, max_len + 1
        sos, eos = self.vocab.str_to_idx(SOS), self.vocab.str_to_idx(EOS)
        key = ("beam", B, K, t_cap)
        if key not in self._buffers:
            self._trim_buffers()
            w0, t0_, shared = self._new_ws(N)
            w1, t1_, _ = self._new_ws(N, shared)
            wss = (L.LstmWorkspace * 2)(w0, w1)
            st, keep = (L.BeamState * 2)(), []
            for i in range(2):
                score = torch.empty(N, dtype=torch.float64, device=dev)
                ln = torch.empty(N, dtype=torch.int32, device=dev)
                seq = torch.empty((N, t_cap), dtype=torch.int32, device=dev)
                keep.append((score, ln, seq))
                st[i].score, st[i].len, st[i].seq, st[i].anc = score.data_ptr(), ln.data_ptr(), seq.data_ptr(), None
            sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(dev)
            self._buffers[key] = (wss, (t0_, t1_), st, keep, torch.empty(N, dtype=torch.int32, device=dev),
                                  torch.empty(t_cap, dtype=torch.int32, device=dev), torch.empty(N, K, dtype=torch.float32, device=dev),
                                  torch.empty(N, K, dtype=torch.int32, device=dev), sqrt_tab)
        wss, tts, st, keep, parent, live, top_val, top_idx, sqrt_tab = self._buffers[key]
        self._init_state(feats, K, wss[0], tts[0])
        stream = L.stream_ptr()
        L.check(L.lib.mic_beam_init(st, live.data_ptr(), N, t_cap, sos, stream), "mic_beam_init")
        steps, s = 0, 0
        while s < max_len:
            e = min(max_len, s + self.sync_every)
            L.check(L.lib.mic_lstm_beam_steps(C.byref(dec), wss, st, parent.data_ptr(), live.data_ptr(), sqrt_tab.data_ptr(),
                                              top_val.data_ptr(), top_idx.data_ptr(), B, K, s, e, t_cap, eos, int(self.use_graph),
                                              stream), "mic_lstm_beam_steps")
            steps = s = e
            if e < max_len and int(live[e - 1].item()) == 0:
                break
        tokens = torch.empty(B, t_cap, dtype=torch.int32, device=dev)
        lengths = torch.empty(B, dtype=torch.int32, device=dev)
        score = torch.empty(B, dtype=torch.float32, device=dev)
        L.check(L.lib.mic_beam_finalize(C.byref(st[steps & 1]), sqrt_tab.data_ptr(), B, K, t_cap, tokens.data_ptr(), lengths.data_ptr(),
                                        score.data_ptr(), None, 0, None, stream), "mic_beam_finalize")
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
