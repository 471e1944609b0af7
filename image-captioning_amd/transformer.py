"""Transformer captioner on libmic_hip: host-side mirror of models/transformer.py:296-640.

`ImageCaptioner.generate` keeps the reference's signature and return convention
(models/transformer.py:394-431): ``(captions, log_probs)`` or ``(captions, log_probs, attn)``.
The algorithm underneath is the cached formulation of the same search: one new position per step
for all images and beams at once, keys/values kept per beam slot with an ancestry table instead of
re-running the prefix, beam bookkeeping on the device in fp64.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib as L
from . import weights as W
from .encoders import ResNet50Encoder, SwinV2Encoder
from .vocabulary import EOS, PAD, SOS, UNK


class ImageCaptioner:
    def __init__(self, vocab, num_heads: int = 2, dtype: torch.dtype = torch.bfloat16, encoder: str = "resnet50",
                 sync_every: int = 8, use_graph: bool = True, decode_groups: int = 1, persistent: bool = True):
        if encoder not in ("resnet50", "swin", "none"):
            raise ValueError(f"Encoder {encoder} not recognized")
        self.vocab = vocab
        self.num_heads = num_heads
        self.dtype = dtype
        self.encoder_kind = encoder
        self.sync_every = max(1, int(sync_every))
        self.use_graph = bool(use_graph)  # replay each run of decode steps as one HIP graph
        # beam search over a large batch runs as this many independent sub-batches on their own streams:
        # the per-step kernels are latency bound, so concurrent sub-batches fill the otherwise idle CUs
        self.decode_groups = max(1, int(decode_groups))
        # one launch per decode step (mic_tr_step_persistent) where the shape allows it.  That launch wants the GPU to
        # itself (its workgroups wait for each other), so it is not used while several searches share the device:
        # concurrent decode groups, or lanes.Lanes with more than one batch in flight (which sets _concurrent).
        self.persistent = bool(persistent)
        self._concurrent = False
        self._ws_lock = threading.Lock()
        self._side_streams = []
        self._sd = None
        self.device = None
        self.encoder = None
        self.dec = None
        self._ws_cache = {}

    # ---- weights ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts ``checkpoint["model_state"]`` of the reference (plotter/captions.py:396)."""
        need = ["seq_embedding.token_embedding.weight", "seq_embedding.pos_embedding.weight", "output_layer.linear.weight",
                "output_layer.linear.bias", "decoder_layers.0.self_attention.mha.in_proj_weight"]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"state dict is missing transformer decoder parameters: {missing}")
        if self.encoder_kind == "resnet50" and "encoder.resnet.0.weight" not in state_dict:
            raise KeyError("state dict has no ResNet-50 encoder (encoder.resnet.*); construct with encoder='none' "
                           "to decode from precomputed features")
        if self.encoder_kind == "swin" and self._swin_prefix(state_dict) is None:
            raise KeyError("state dict has no Swin encoder (encoder.features.* / encoder.swin.features.*)")
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        self.max_len = self._sd["seq_embedding.pos_embedding.weight"].shape[0]
        self.device = None
        return self

    @staticmethod
    def _swin_prefix(sd):
        """models/encoders/swin.py:26,29 registers the trunk twice; either name serves."""
        for p in ("encoder.features.", "encoder.swin.features."):
            if p + "0.0.weight" in sd:
                return p
        return None

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("image_captioning_amd runs on a HIP device only; there is no CPU path")
        if self._sd is None:
            raise RuntimeError("load_state_dict() must be called before to(device)")
        if self.device == device:
            return self
        with torch.cuda.device(device):
            self.dec, self._dec_keep = W.build_tr_decoder(self._sd, self.num_heads, self.dtype, device)
            if self.encoder_kind == "resnet50":
                self.encoder = ResNet50Encoder(self._sd, "transformer", self.dtype, device)
            elif self.encoder_kind == "swin":
                self.encoder = SwinV2Encoder(self._sd, self.dtype, device, self._swin_prefix(self._sd))
            else:
                self.encoder = None
        self.device = device
        self._ws_cache = {}
        return self

    def eval(self):
        return self

    def lane_copy(self):
        """A second handle on the same device weights with workspaces of its own, for a concurrent lane (lanes.Lanes)."""
        import copy
        if self.device is None:
            raise RuntimeError("lane_copy() needs a model that is on its device")
        m = copy.copy(self)
        m._ws_cache, m._side_streams, m._ws_lock = {}, [], threading.Lock()
        if self.encoder is not None:
            m.encoder = copy.copy(self.encoder)
            m.encoder._ws = None
        return m

    # ---- workspace ----------------------------------------------------------------------------
    def _workspace(self, N: int, t_cap: int, S: int, rpi: int, K: int, attn_mode: int, slot: int = 0):
        want_persist = self.persistent and not self._concurrent and self.decode_groups == 1
        key = (N, t_cap, S, rpi, K, attn_mode, slot, want_persist)
        with self._ws_lock:  # decode groups and lanes call from several host threads
            hit = self._ws_cache.pop(key, None)
            if hit is not None:
                self._ws_cache[key] = hit  # most recently used last
                return hit
        d, dev, H, V, Ly = self.dtype, self.device, self.dec.H, self.dec.vocab, self.dec.layers
        f32 = torch.float32
        t = {
            "x": torch.empty(N, H, dtype=f32, device=dev), "xt": torch.empty(N, H, dtype=d, device=dev),
            "y": torch.empty(N, H, dtype=f32, device=dev), "qkv": torch.empty(N, 3 * H, dtype=f32, device=dev),
            "o": torch.empty(N, H, dtype=d, device=dev), "ffh": torch.empty(N, 2 * H, dtype=d, device=dev),
            "logits": torch.empty(N, V, dtype=f32, device=dev),
            "kc": torch.zeros(Ly, N, t_cap, H, dtype=d, device=dev), "vc": torch.zeros(Ly, N, t_cap, H, dtype=d, device=dev),
            "ckv": torch.empty(Ly, N // rpi, S, 2 * H, dtype=d, device=dev),
            "top_val": torch.empty(N, K, dtype=f32, device=dev), "top_idx": torch.empty(N, K, dtype=torch.int32, device=dev),
            "stats": torch.zeros(2, N, (H + 127) // 128, 2, dtype=f32, device=dev),
        }
        heads = self.dec.heads
        fold = bool(L.lib.mic_cross_attn_folded_supported(L.dtype_code(d), H, heads, S, rpi)) and bool(self.dec.layer[0].cross_qb_scaled)
        if fold:  # folded cross keys / values (query and output projections applied once per image)
            t["ckf"] = torch.empty(Ly, N // rpi, S, heads, H, dtype=d, device=dev)
            t["cvf"] = torch.empty(Ly, N // rpi, S, heads, H, dtype=d, device=dev)
            t["cc"] = torch.empty(Ly, N // rpi, S, heads, dtype=f32, device=dev)
        fuse = fold and bool(self.dec.layer[0].self_qkvf_wf) and bool(
            L.lib.mic_attn_fused_supported(L.dtype_code(d), H, heads, S, rpi, t_cap))
        if fuse:  # both attentions of a layer in one launch: the cached values carry the output projection
            t["qkvf"] = torch.empty(N, (2 + heads) * H, dtype=f32, device=dev)
            t["vfc"] = torch.zeros(Ly, N, t_cap, heads, H, dtype=d, device=dev)
        persist = fuse and want_persist and V % 4 == 0 and bool(
            L.lib.mic_tr_persist_supported(L.dtype_code(d), H, heads, Ly, S, rpi, t_cap, N))
        if persist:  # seam counters and the error word of the one-launch step
            t["sync"] = torch.zeros(int(L.lib.mic_tr_persist_sync_words(N, rpi)), dtype=torch.int32, device=dev)
        if attn_mode == 1:
            t["attn"] = torch.zeros(N, t_cap, S, dtype=f32, device=dev)
        elif attn_mode == 2:
            t["attn"] = torch.zeros(Ly, t_cap, N, S, dtype=f32, device=dev)
        ws = L.TrWorkspace()
        ws.N, ws.t_cap, ws.S, ws.rows_per_image = N, t_cap, S, rpi
        for name in ("x", "xt", "y", "qkv", "o", "ffh", "logits", "top_val", "top_idx"):
            setattr(ws, name, t[name].data_ptr())
        for l in range(Ly):
            ws.kcache[l], ws.vcache[l], ws.cross_kv[l] = t["kc"][l].data_ptr(), t["vc"][l].data_ptr(), t["ckv"][l].data_ptr()
            if fold:
                ws.cross_kf[l], ws.cross_vf[l], ws.cross_c[l] = t["ckf"][l].data_ptr(), t["cvf"][l].data_ptr(), t["cc"][l].data_ptr()
        if fuse:
            ws.qkvf = t["qkvf"].data_ptr()
            for l in range(Ly):
                ws.vfcache[l] = t["vfc"][l].data_ptr()
        ws.attn = t["attn"].data_ptr() if attn_mode else None
        ws.stats[0], ws.stats[1] = t["stats"][0].data_ptr(), t["stats"][1].data_ptr()
        ws.sync = t["sync"].data_ptr() if persist else None
        with self._ws_lock:
            # at most four workspaces per slot, the least recently used of THIS slot dropped first: another group's
            # workspace may be in the middle of a search
            mine = [k for k in self._ws_cache if k[6] == slot]
            for k in mine[: max(0, len(mine) - 3)]:
                del self._ws_cache[k]
            self._ws_cache[key] = (ws, t)
        return ws, t

    def step_schedule(self, ws) -> str:
        """Name of the schedule a decode step on this workspace takes inside the search drivers."""
        return {3: "persistent", 2: "fused14", 1: "chained", 0: "tiled"}.get(
            int(L.lib.mic_tr_step_schedule(C.byref(self.dec), C.byref(ws))), "unknown")

    def check_errors(self):
        """Raises if a one-launch decode step of any cached workspace reported a timed-out wait (synchronises)."""
        with self._ws_lock:
            held = [t for _, t in self._ws_cache.values()]
        for t in held:
            self._check_sync(t)

    @staticmethod
    def _check_sync(t):
        """The one-launch step reports a wait that ran out in its error word instead of hanging the GPU."""
        if "sync" in t:
            code = int(t["sync"][-64].item())
            if code:
                t["sync"].zero_()
                raise RuntimeError(f"persistent decode step: a wait between workgroups timed out (code {code:#x}); "
                                   "was another kernel occupying the GPU?")

    # ---- search -------------------------------------------------------------------------------
    def _check_feats(self, feats: torch.Tensor) -> torch.Tensor:
        if feats.dim() != 3 or feats.shape[2] != self.dec.H:
            raise ValueError(f"expected image tokens of shape (B,S,{self.dec.H}), got {tuple(feats.shape)}")
        return feats.to(self.device, self.dtype).contiguous()

    def beam_search_ids(self, feats: torch.Tensor, max_len: int, beam_size: int, return_attn: bool = False):
        """Device beam search (models/transformer.py:576-640).  Returns int32 tokens (B, max_len+1) padded
        with -1, int32 lengths (B,), fp32 summed log-probs (B,), and fp32 attention (B, max_len+1, S) or None."""
        if not 1 < beam_size <= L.MAX_BEAM:
            raise ValueError(f"beam_size must be in [2, {L.MAX_BEAM}], got {beam_size}")
        feats = self._check_feats(feats)
        B, S, _ = feats.shape
        dev, t_cap = self.device, max_len + 1
        tokens = torch.empty(B, t_cap, dtype=torch.int32, device=dev)
        lengths = torch.empty(B, dtype=torch.int32, device=dev)
        score = torch.empty(B, dtype=torch.float32, device=dev)
        attn = torch.zeros(B, t_cap, S, dtype=torch.float32, device=dev) if return_attn else None
        G = min(self.decode_groups, B)
        if G == 1:
            self._beam_group(feats, max_len, beam_size, tokens, lengths, score, attn, 0)
            return tokens, lengths, score, attn
        # group 0 runs on the caller's thread and stream, every other group on a host thread and stream of its own: a
        # run of decode steps is enqueued at about the rate the GPU consumes it, so groups enqueued one after another
        # by one thread would also run one after another
        while len(self._side_streams) < G - 1:
            self._side_streams.append(torch.cuda.Stream(device=dev))
        main = torch.cuda.current_stream(dev)
        ready = main.record_event()
        bounds = [(B * g) // G for g in range(G + 1)]
        errors, done = [], [None] * G

        def work(g):
            lo, hi = bounds[g], bounds[g + 1]
            try:
                if g == 0:
                    self._beam_group(feats[lo:hi], max_len, beam_size, tokens[lo:hi], lengths[lo:hi], score[lo:hi],
                                     attn[lo:hi] if return_attn else None, g)
                    return
                torch.cuda.set_device(dev)
                side = self._side_streams[g - 1]
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    self._beam_group(feats[lo:hi], max_len, beam_size, tokens[lo:hi], lengths[lo:hi], score[lo:hi],
                                     attn[lo:hi] if return_attn else None, g)
                    done[g] = side.record_event()
            except BaseException as e:  # re-raised on the calling thread
                errors.append(e)

        threads = [threading.Thread(target=work, args=(g,)) for g in range(1, G)]
        for th in threads:
            th.start()
        work(0)
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        for ev in done[1:]:
            main.wait_event(ev)
        return tokens, lengths, score, attn

    def _beam_group(self, feats, max_len, beam_size, tokens, lengths, score, attn, slot):
        """Beam search of one contiguous sub-batch on the current stream; results go into the given row slices."""
        return_attn = attn is not None
        B, S, _ = feats.shape
        K, dev = beam_size, self.device
        N, t_cap = B * K, max_len + 1
        ws, t = self._workspace(N, t_cap, S, K, K, 1 if return_attn else 0, slot)
        sos = self.vocab.str_to_idx(SOS)
        if "beam" not in t:  # search state lives with the workspace so that repeated calls reuse addresses
            st = (L.BeamState * 2)()
            bufs = []
            for i in range(2):
                b = (torch.empty(N, dtype=torch.float64, device=dev), torch.empty(N, dtype=torch.int32, device=dev),
                     torch.empty(N, t_cap, dtype=torch.int32, device=dev), torch.empty(N, t_cap, dtype=torch.int32, device=dev))
                bufs.append(b)
                st[i].score, st[i].len, st[i].seq, st[i].anc = (x.data_ptr() for x in b)
            # len ** 0.5 exactly as the reference's Python evaluates it (models/transformer.py:627)
            sqrt_tab = torch.tensor([float(i) ** 0.5 for i in range(t_cap + 2)], dtype=torch.float64).to(dev)
            t["beam"] = (st, bufs, torch.empty(N, dtype=torch.int32, device=dev), torch.empty(max(t_cap, 1), dtype=torch.int32, device=dev),
                         sqrt_tab)
        st, bufs, parent, live, sqrt_tab = t["beam"]
        stream = L.stream_ptr()
        L.check(L.lib.mic_beam_init(st, live.data_ptr(), N, t_cap, sos, stream), "mic_beam_init")
        L.check(L.lib.mic_tr_cross_kv(C.byref(self.dec), C.byref(ws), feats.data_ptr(), stream), "mic_tr_cross_kv")
        eos = self.vocab.str_to_idx(EOS)
        t0 = 0
        while t0 < max_len:
            t1 = min(max_len, t0 + self.sync_every)
            L.check(L.lib.mic_tr_beam_steps(C.byref(self.dec), C.byref(ws), st, parent.data_ptr(), live.data_ptr(),
                                            sqrt_tab.data_ptr(), B, K, t0, t1, eos, int(return_attn), int(self.use_graph), stream),
                    "mic_tr_beam_steps")
            t0 = t1
            if t0 < max_len and int(live[t0 - 1].item()) == 0:  # every image's beams have ended
                break
        L.check(L.lib.mic_beam_finalize(C.byref(st[t0 & 1]), sqrt_tab.data_ptr(), B, K, t_cap, tokens.data_ptr(), lengths.data_ptr(),
                                        score.data_ptr(), t["attn"].data_ptr() if return_attn else None, S,
                                        L.ptr(attn), stream), "mic_beam_finalize")
        self.last_schedule = self.step_schedule(ws)

    def sample_ids(self, feats: torch.Tensor, max_len: int, temp: Optional[float], return_attn: bool = False,
                   uniforms: Optional[torch.Tensor] = None):
        """Batched greedy / temperature sampling (models/transformer.py:475-556).  Returns int32 tokens
        (B, n_tokens), fp32 log-probs (B,), attention (L, steps, B, S) or None."""
        feats = self._check_feats(feats)
        B, S, _ = feats.shape
        dev = self.device
        hot = temp is not None and temp > 0
        t_cap = max(max_len, 1)
        ws, t = self._workspace(B, t_cap, S, 1, 1, 2 if return_attn else 0)
        if "samp" not in t:
            t["samp"] = (torch.empty(B, t_cap, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.uint8, device=dev),
                         torch.empty(B, dtype=torch.float32, device=dev), torch.empty(t_cap, dtype=torch.int32, device=dev),
                         torch.empty(max(t_cap - 1, 1), B, dtype=torch.float32, device=dev))
        tokens, finished, total, live, ubuf = t["samp"]
        tokens.fill_(self.vocab.str_to_idx(PAD))
        tokens[:, 0] = self.vocab.str_to_idx(SOS)
        finished.zero_(); total.zero_(); live.zero_()
        steps = max_len - 1
        u = None
        if hot and steps > 0:
            if uniforms is None:
                ubuf.uniform_()
            else:
                if uniforms.shape[0] < steps or uniforms.shape[1] != B:
                    raise ValueError(f"uniforms must have shape (>={steps}, {B})")
                ubuf[:steps].copy_(uniforms[:steps].to(dev, torch.float32))
            u = ubuf
        stream = L.stream_ptr()
        L.check(L.lib.mic_tr_cross_kv(C.byref(self.dec), C.byref(ws), feats.data_ptr(), stream), "mic_tr_cross_kv")
        t0, done_steps = 0, steps
        while t0 < steps:
            t1 = min(steps, t0 + self.sync_every)
            L.check(L.lib.mic_tr_sample_steps(C.byref(self.dec), C.byref(ws), tokens.data_ptr(), finished.data_ptr(), total.data_ptr(),
                                              live.data_ptr(), (1.0 / temp) if hot else 1.0, L.ptr(u), t0, t1,
                                              self.vocab.str_to_idx(PAD), self.vocab.str_to_idx(EOS), int(return_attn),
                                              int(self.use_graph), stream),
                    "mic_tr_sample_steps")
            lv = live[t0:t1].tolist()
            t0 = t1
            if 0 in lv:  # the reference stops right after the step in which the last row finished
                done_steps = t1 - len(lv) + lv.index(0) + 1
                break
        attn = t["attn"][:, :done_steps].clone() if return_attn else None
        self.last_schedule = self.step_schedule(ws)
        return tokens[:, : done_steps + 1].clone(), total.clone(), attn

    # ---- reference-compatible API -------------------------------------------------------------
    def generate(self, images: torch.Tensor, vocab, max_len: int, device, temp: Optional[float], beam_size: int,
                 no_grad: bool = True, return_attn: bool = False, **kwargs) -> tuple:
        """models/transformer.py:394-431.  ``no_grad=False`` (used only by the reference's RL training
        loop, train.py:235) would need autograd through the kernels and is rejected."""
        if "return_atnn" in kwargs:  # the spelling captioner.py:30 uses
            return_attn = kwargs.pop("return_atnn")
        if kwargs:
            raise TypeError(f"generate() got unexpected keyword arguments {sorted(kwargs)}")
        if not no_grad:
            raise NotImplementedError("generate(no_grad=False) is training-only and not supported by the HIP path")
        self.to(device)
        eff = min(max_len, self.max_len)
        feats = self.encode(images)
        return self.generate_from_features(feats, vocab, eff, temp, beam_size, return_attn)

    def encode(self, images: torch.Tensor) -> torch.Tensor:
        if self.encoder is None:
            raise RuntimeError("this captioner was built without an encoder; pass features to generate_from_features")
        return self.encoder(images)

    # ---- teacher-forced evaluation (no autograd: inference-side loss only) -----------------------
    def forward_from_features(self, feats: torch.Tensor, captions: torch.Tensor, want_lse: bool = False):
        """models/transformer.py:338-363 downstream of the encoder: next-token logits (B, T, V) fp32 for every
        position of ``captions`` (B, T), computed by stepping the cached decoder over the given tokens (position p
        sees tokens 0..p, exactly what the causal mask lets the reference see).  With ``want_lse`` also returns
        log-sum-exp of every row (B, T), from the library's log-softmax kernel."""
        feats = self._check_feats(feats)
        B, S, _ = feats.shape
        if captions.dim() != 2 or captions.shape[0] != B:
            raise ValueError(f"expected captions of shape ({B}, T), got {tuple(captions.shape)}")
        T = captions.shape[1]
        if T < 1 or T > self.max_len:
            raise ValueError(f"caption length {T} outside [1, {self.max_len}] (position table size)")
        dev, V = self.device, self.dec.vocab
        cap = captions.to(dev, torch.int32).contiguous()
        ws, t = self._workspace(B, T, S, 1, 1, 0)
        out = torch.empty(B, T, V, dtype=torch.float32, device=dev)
        lse = torch.empty(T, B, dtype=torch.float32, device=dev) if want_lse else None
        stream = L.stream_ptr()
        L.check(L.lib.mic_tr_cross_kv(C.byref(self.dec), C.byref(ws), feats.data_ptr(), stream), "mic_tr_cross_kv")
        for p in range(T):
            L.check(L.lib.mic_tr_step(C.byref(self.dec), C.byref(ws), cap.data_ptr(), T, None, p, 0, stream), "mic_tr_step")
            if want_lse:
                L.check(L.lib.mic_logsoftmax_topk(t["logits"].data_ptr(), V, B, V, 1, 1.0, 0, t["top_val"].data_ptr(),
                                                  t["top_idx"].data_ptr(), lse[p].data_ptr(), stream), "mic_logsoftmax_topk")
            out[:, p].copy_(t["logits"])
        return (out, lse.t().contiguous()) if want_lse else out

    def forward(self, images: torch.Tensor, captions: torch.Tensor) -> torch.Tensor:
        """models/transformer.py:338-363 (eval mode; the training path with autograd is out of scope)."""
        if self.device is None:
            raise RuntimeError("call .to(device) first")
        return self.forward_from_features(self.encode(images), captions)

    __call__ = forward

    def evaluation_loss(self, images: torch.Tensor, captions: torch.Tensor, vocab=None, feats: Optional[torch.Tensor] = None):
        """train.forward_pass (train.py:318-354) without gradients: summed cross-entropy of captions[:, 1:] given
        captions[:, :-1], over targets other than PAD / SOS / UNK, and the number of such targets."""
        vocab = vocab if vocab is not None else self.vocab
        feats = self.encode(images) if feats is None else feats
        logits, lse = self.forward_from_features(feats, captions[:, :-1], want_lse=True)
        targets = captions[:, 1:].to(self.device, torch.int64)
        picked = logits.gather(2, targets.clamp(0, self.dec.vocab - 1).unsqueeze(-1)).squeeze(-1)
        mask = torch.ones_like(targets, dtype=torch.bool)
        for tok in (PAD, SOS, UNK):
            mask &= targets != vocab.str_to_idx(tok)
        n = int(mask.sum().item())
        loss = ((lse - picked) * mask).sum() if n > 0 else torch.zeros((), device=self.device)
        return loss, n

    def generate_from_features(self, feats, vocab, max_len: int, temp, beam_size: int, return_attn: bool = False,
                               uniforms=None) -> tuple:
        """Everything downstream of the encoder (models/transformer.py:452-456)."""
        max_len = min(max_len, self.max_len)
        if beam_size > 1:
            tokens, lengths, score, attn = self.beam_search_ids(feats, max_len, beam_size, return_attn)
            tok, ln = tokens.cpu().numpy(), lengths.cpu().numpy()
            self.check_errors()
            ids = [tok[b, : ln[b]].tolist() for b in range(tok.shape[0])]
            captions = [vocab.encode_as_words(i) for i in ids]
            if not return_attn:
                return captions, score
            a = attn.cpu().numpy()
            return captions, score, [[a[b, s] for s in range(ln[b] - 1)] for b in range(len(ids))]
        tokens, total, attn = self.sample_ids(feats, max_len, temp, return_attn, uniforms)
        captions = [vocab.encode_as_words(row) for row in tokens.cpu().numpy().tolist()]
        self.check_errors()
        if not return_attn:
            return captions, total
        a = attn.cpu().numpy()
        return captions, total, [[a[l, s] for s in range(a.shape[1])] for l in range(a.shape[0])]

    @classmethod
    def from_state_dict(cls, state_dict, vocab, num_heads: int = 2, dtype=torch.bfloat16, device="cuda",
                        encoder: Optional[str] = None, **kw) -> "ImageCaptioner":
        if encoder is None:
            encoder = ("resnet50" if "encoder.resnet.0.weight" in state_dict
                       else "swin" if cls._swin_prefix(state_dict) else "none")
        return cls(vocab, num_heads, dtype, encoder, **kw).load_state_dict(state_dict).to(device)
