"""ctypes binding of libmic_hip.so (declared in include/mic_hip.h).

The library is the product; there is no fallback.  Importing this module loads the shared object
built by ``csrc/build.sh`` and raises if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os

F32, BF16 = 0, 1
MAX_BEAM, MAX_LAYERS = 8, 8
STATIC_SCRATCH_BYTES = 8 << 20  # MIC_STATIC_SCRATCH_BYTES

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmic_hip.so")

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("in_", vp), ("w", vp), ("bias", vp), ("res", vp), ("out", vp),
                ("dtype", i32), ("out_dtype", i32), ("res_dtype", i32),
                ("M", i32), ("N", i32), ("K", i32),
                ("OH", i32), ("OW", i32), ("IH", i32), ("IW", i32),
                ("in_img_stride", i64), ("in_row_stride", i64),
                ("in_pix_stride", i32), ("kw_taps", i32), ("run", i32),
                ("stride", i32), ("pad_h", i32), ("pad_w", i32),
                ("ldc", i32), ("ld_res", i32), ("relu", i32), ("k_valid", i32),
                ("in2", vp), ("in2_img_stride", i64), ("in2_row_stride", i64),
                ("in2_pix_stride", i32), ("stride2", i32), ("K2", i32), ("w_rows", i32),
                ("ln_gamma", vp), ("ln_beta", vp), ("ln_out", vp), ("ld_ln_out", i32),
                ("scratch", vp), ("scratch_bytes", i64)]


class FusedBlockDesc(C.Structure):
    _fields_ = [("t1", vp), ("w2", vp), ("b2", vp), ("w3", vp), ("b3", vp), ("res", vp), ("xout", vp), ("w1n", vp), ("b1n", vp),
                ("t1n", vp), ("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("C2", i32),
                ("w2_rows", i32), ("w3_rows", i32), ("w1n_rows", i32), ("xs", vp), ("CS", i32)]


class LnRef(C.Structure):
    _fields_ = [("raw", vp), ("stats", vp), ("gamma", vp), ("beta", vp), ("ld", i32), ("parts", i32), ("width", i32)]


class SkinnyDesc(C.Structure):
    _fields_ = [("x", vp), ("x_ln", LnRef), ("wf", vp), ("bias", vp), ("res", vp), ("res_ln", LnRef), ("out", vp), ("stats_out", vp),
                ("dtype", i32), ("out_dtype", i32), ("ld_x", i32), ("ld_res", i32), ("ldc", i32), ("M", i32), ("N", i32), ("K", i32),
                ("relu", i32)]


GATHER_MAX = 16


class BeamTailDesc(C.Structure):
    _fields_ = [("logits", vp), ("ld", i32), ("V", i32), ("cur", vp), ("nxt", vp), ("parent", vp), ("live_images", vp), ("sqrt_tab", vp),
                ("B", i32), ("K", i32), ("n_live_slots", i32), ("t", i32), ("t_cap", i32), ("eos_id", i32), ("finished_in_place", i32),
                ("tok_emb", vp), ("pos_emb", vp), ("x", vp), ("xt", vp), ("ld_xt", i32), ("dtype", i32), ("H", i32), ("vocab", i32),
                ("n_gather", i32), ("gather_src", vp * GATHER_MAX), ("gather_dst", vp * GATHER_MAX),
                ("gather_row_bytes", i32 * GATHER_MAX), ("gather_stride_bytes", i32 * GATHER_MAX), ("row_group_threads", i32)]


class AttnFusedDesc(C.Structure):
    _fields_ = [("qkvf", vp), ("kcache", vp), ("vfcache", vp), ("anc", vp), ("x", vp), ("x_ln", LnRef), ("self_b", vp), ("ln1_g", vp),
                ("ln1_b", vp), ("kf", vp), ("vf", vp), ("c", vp), ("bo", vp), ("y", vp), ("stats_out", vp), ("attn_out", vp),
                ("ld_attn", i32), ("accumulate", i32), ("attn_scale", f32), ("dtype", i32), ("N", i32), ("H", i32), ("heads", i32),
                ("S", i32), ("rows_per_image", i32), ("t", i32), ("t_cap", i32)]


class BeamState(C.Structure):
    _fields_ = [("score", vp), ("len", vp), ("seq", vp), ("anc", vp)]


class ConvLayer(C.Structure):
    _fields_ = [("w", vp), ("bias", vp), ("cin", i32), ("cout", i32), ("kh", i32), ("kw", i32),
                ("stride", i32), ("pad", i32), ("relu", i32)]


class FusedTail(C.Structure):
    _fields_ = [("w", vp), ("bias", vp), ("cin_main", i32), ("cin_skip", i32), ("cout", i32), ("skip_stride", i32)]


class Resnet50(C.Structure):
    _fields_ = [("dtype", i32), ("stem", ConvLayer), ("blocks_per_stage", i32 * 4), ("conv", (ConvLayer * 4) * 16),
                ("tail", FusedTail * 4)]


SWIN_MAX_BLOCKS = 32


class SwinBlock(C.Structure):
    _fields_ = [(n, vp) for n in ["qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_b", "proj_b", "fc1_b", "fc2_b",
                                  "n1_g", "n1_b", "n2_g", "n2_b", "bias_tab", "logit_scale"]]


class SwinMerge(C.Structure):
    _fields_ = [("w", vp), ("g", vp), ("b", vp)]


class Swin(C.Structure):
    _fields_ = [("dtype", i32), ("embed_dim", i32), ("depths", i32 * 4), ("heads", i32 * 4),
                ("patch_w", vp), ("patch_b", vp), ("patch_g", vp), ("patch_be", vp),
                ("block", SwinBlock * SWIN_MAX_BLOCKS), ("merge", SwinMerge * 3)]


_TR_W = ["self_in_w", "self_out_w", "cross_q_w", "cross_kv_w", "cross_out_w", "ff1_w", "ff2_w"]
_TR_B = ["self_in_b", "self_out_b", "cross_q_b", "cross_kv_b", "cross_out_b", "ff1_b", "ff2_b"]
_TR_LN = ["ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b"]


FOLD_MAX_HEADS = 2


class TrLayer(C.Structure):
    _fields_ = ([(n, vp) for n in _TR_W + _TR_B + _TR_LN + ["self_out_wf", "cross_out_wf", "ff2_wf", "self_in_wf", "cross_q_wf", "ff1_wf"]]
                + [("cross_kf_w", vp * FOLD_MAX_HEADS), ("cross_vf_w", vp * FOLD_MAX_HEADS), ("cross_qb_scaled", vp),
                   ("cross_kfc_w", vp), ("cross_vfc_w", vp), ("cross_kfc_b", vp), ("cross_vfc_b", vp), ("cross_cu", vp), ("cross_c0", vp),
                   ("self_qkvf_wf", vp), ("self_qkvf_b", vp), ("self_of_b", vp)])


class TrDecoder(C.Structure):
    _fields_ = [("dtype", i32), ("H", i32), ("heads", i32), ("layers", i32), ("vocab", i32), ("pos_rows", i32),
                ("tok_emb", vp), ("pos_emb", vp), ("layer", TrLayer * MAX_LAYERS), ("out_w", vp), ("out_b", vp), ("out_wf", vp)]


class TrWorkspace(C.Structure):
    _fields_ = [("N", i32), ("t_cap", i32), ("S", i32), ("rows_per_image", i32),
                ("x", vp), ("xt", vp), ("y", vp), ("qkv", vp), ("o", vp), ("ffh", vp), ("logits", vp),
                ("kcache", vp * MAX_LAYERS), ("vcache", vp * MAX_LAYERS), ("cross_kv", vp * MAX_LAYERS),
                ("attn", vp), ("top_val", vp), ("top_idx", vp), ("stats", vp * 2),
                ("cross_kf", vp * MAX_LAYERS), ("cross_vf", vp * MAX_LAYERS), ("cross_c", vp * MAX_LAYERS),
                ("qkvf", vp), ("vfcache", vp * MAX_LAYERS), ("sync", vp)]


class LstmDecoder(C.Structure):
    _fields_ = [("dtype", i32), ("E", i32), ("Hd", i32), ("layers", i32), ("vocab", i32), ("kind", i32),
                ("kpad", i32 * MAX_LAYERS), ("embed", vp), ("w", vp * MAX_LAYERS), ("b", vp * MAX_LAYERS),
                ("ln_g", vp), ("ln_b", vp), ("out_w", vp), ("out_b", vp), ("out_k", i32),
                ("wf", vp * MAX_LAYERS), ("out_wf", vp), ("wf_cell", vp * MAX_LAYERS), ("b_cell", vp * MAX_LAYERS)]


class LstmWorkspace(C.Structure):
    _fields_ = [("N", i32), ("xh", vp * MAX_LAYERS), ("h", vp * MAX_LAYERS), ("c", vp * MAX_LAYERS),
                ("gates", vp), ("hn", vp), ("hnt", vp), ("logits", vp), ("scratch", vp), ("hstage", vp * MAX_LAYERS)]


P = C.POINTER
# name -> (restype, argtypes); must list every symbol include/mic_hip.h declares
SIGNATURES = {
    "mic_last_error": (C.c_char_p, []),
    "mic_version": (C.c_int, []),
    "mic_conv2d": (C.c_int, [P(ConvDesc), vp]),
    "mic_conv2d_variant": (C.c_int, [P(ConvDesc), C.c_int, vp]),
    "mic_conv2d_ln_supported": (C.c_int, [C.c_int]),
    "mic_conv2d_scratch_bytes": (C.c_int64, []),
    "mic_bottleneck_fused_shortcut_supported": (C.c_int, [C.c_int] * 4),
    "mic_stem_pool_supported": (C.c_int, [C.c_int]),
    "mic_stem_pool": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 7 + [vp]),
    "mic_stem_pool_conv1": (C.c_int, [vp] * 7 + [C.c_int] * 8 + [vp]),
    "mic_bottleneck_fused_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mic_bottleneck_fused": (C.c_int, [P(FusedBlockDesc), vp]),
    "mic_profile_begin": (C.c_int, [C.c_int]),
    "mic_profile_end": (C.c_int, [P(i32), P(C.c_double), P(C.c_double)]),
    "mic_pack_image": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_resize_normalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(f32), P(f32), C.c_int, vp]),
    "mic_maxpool3x3s2": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_global_avgpool": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_embed_pos": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_layernorm": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_layernorm_add": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_patchify4": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_window_attn": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_swin_workspace_bytes": (C.c_int64, [P(Swin), C.c_int, C.c_int, C.c_int]),
    "mic_swin_forward": (C.c_int, [P(Swin), vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, vp]),
    "mic_pack_weight_frag": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mic_linear_skinny": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_linear_skinny_ex": (C.c_int, [C.POINTER(SkinnyDesc), vp]),
    "mic_lstm_gates_cell": (C.c_int, [C.POINTER(SkinnyDesc), vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "mic_linear_res_ln_supported": (C.c_int, [C.c_int]),
    "mic_linear_res_ln": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_self_attn_step": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_cross_attn_step": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, f32, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_cross_attn_folded_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mic_cross_attn_folded": (C.c_int, [P(LnRef), vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, f32, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, vp]),
    "mic_cross_fold_bias": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_cross_fold_c": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_attn_fused_supported": (C.c_int, [C.c_int] * 6),
    "mic_attn_fused": (C.c_int, [P(AttnFusedDesc), vp]),
    "mic_beam_tail_supported": (C.c_int, [C.c_int, C.c_int]),
    "mic_beam_tail": (C.c_int, [vp, C.c_int, C.c_int, P(BeamState), P(BeamState), vp, vp, vp] + [C.c_int] * 6 + [vp, vp, vp, vp]
                      + [C.c_int] * 3 + [vp]),
    "mic_beam_tail_ex": (C.c_int, [P(BeamTailDesc), vp]),
    "mic_beam_init": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mic_lstm_cell": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_lstm_cell_ln_supported": (C.c_int, [C.c_int]),
    "mic_lstm_cell_ln": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_gather_rows": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "mic_gather_rows_multi": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, vp]),
    "mic_logsoftmax_topk": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.c_int, vp, vp, vp, vp]),
    "mic_sample_multinomial": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, vp]),
    "mic_beam_select": (C.c_int, [P(BeamState), P(BeamState), vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, vp]),
    "mic_beam_finalize": (C.c_int, [P(BeamState), vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]),
    "mic_sample_update": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mic_resnet50_workspace_bytes": (C.c_int64, [P(Resnet50), C.c_int, C.c_int, C.c_int]),
    "mic_resnet50_forward": (C.c_int, [P(Resnet50), vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, vp]),
    "mic_tr_cross_kv": (C.c_int, [P(TrDecoder), P(TrWorkspace), vp, vp]),
    "mic_tr_step": (C.c_int, [P(TrDecoder), P(TrWorkspace), vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "mic_tr_step_persistent": (C.c_int, [P(TrDecoder), P(TrWorkspace), vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "mic_tr_persist_supported": (C.c_int, [C.c_int] * 8),
    "mic_tr_persist_trace": (C.c_int, [vp]),
    "mic_tr_step_schedule": (C.c_int, [P(TrDecoder), P(TrWorkspace)]),
    "mic_tr_persist_sync_words": (C.c_int64, [C.c_int, C.c_int]),
    "mic_tr_step_static": (C.c_int, [P(TrDecoder), P(TrWorkspace), vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "mic_tr_beam_steps": (C.c_int, [P(TrDecoder), P(TrWorkspace), P(BeamState), vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_tr_sample_steps": (C.c_int, [P(TrDecoder), P(TrWorkspace), vp, vp, vp, vp, f32, vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, vp]),
    "mic_graph_cache_stats": (C.c_int, [P(i32), P(i32)]),
    "mic_lstm_step": (C.c_int, [P(LstmDecoder), P(LstmWorkspace), vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "mic_lstm_beam_steps": (C.c_int, [P(LstmDecoder), P(LstmWorkspace), P(BeamState), vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mic_lstm_sample_steps": (C.c_int, [P(LstmDecoder), P(LstmWorkspace), vp, vp, vp, f32, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, vp]),
}


class MicError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with image-captioning_amd/csrc/build.sh "
            "(there is no CPU or PyTorch fallback for the generate path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    """Turn a non-zero return code into an exception carrying mic_last_error()."""
    if rc != 0:
        msg = lib.mic_last_error().decode("utf-8", "replace")
        kind = "argument error" if rc < 0 else f"hipError {rc}"
        raise MicError(f"{what or 'libmic_hip'}: {kind}: {msg}")


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def dtype_code(torch_dtype) -> int:
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise ValueError(f"unsupported compute dtype {torch_dtype}; use torch.float32 or torch.bfloat16")
