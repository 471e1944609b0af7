"""The C-ABI library loads without a GPU and exports exactly what include/mic_hip.h declares;
argument validation that happens before any HIP call is exercised here too."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mic_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mic_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import image_captioning_amd as amd
    L = amd._lib
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L.lib, s), f"{s} declared in mic_hip.h but not exported by libmic_hip.so"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(L.SIGNATURES) == syms, "ctypes table and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("mic_")}
    assert exported == set(syms), f"exported-but-undeclared or missing: {exported ^ set(syms)}"


def test_struct_layouts_match_the_header():
    """Compile a tiny C program against the header and compare sizeof / a few offsets with ctypes."""
    import image_captioning_amd as amd
    L = amd._lib
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "mic_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mic_conv_desc), sizeof(mic_beam_state), sizeof(mic_conv_layer),
             sizeof(mic_resnet50), sizeof(mic_tr_layer), sizeof(mic_tr_decoder), sizeof(mic_tr_workspace),
             sizeof(mic_lstm_decoder), sizeof(mic_lstm_workspace), sizeof(mic_ln_ref), sizeof(mic_skinny_desc),
             sizeof(mic_swin_block), sizeof(mic_swin_merge), sizeof(mic_swin), sizeof(mic_fused_tail), sizeof(mic_attn_fused_desc), sizeof(mic_beam_tail_desc));
      printf("%zu %zu %zu %zu\n", offsetof(mic_conv_desc, in_img_stride), offsetof(mic_conv_desc, k_valid),
             offsetof(mic_tr_decoder, out_w), offsetof(mic_lstm_decoder, out_k));
      return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes, offs = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    got = [C.sizeof(x) for x in (L.ConvDesc, L.BeamState, L.ConvLayer, L.Resnet50, L.TrLayer, L.TrDecoder, L.TrWorkspace,
                                 L.LstmDecoder, L.LstmWorkspace, L.LnRef, L.SkinnyDesc, L.SwinBlock, L.SwinMerge, L.Swin, L.FusedTail, L.AttnFusedDesc, L.BeamTailDesc)]
    assert got == [int(v) for v in sizes.split()]
    assert [L.ConvDesc.in_img_stride.offset, L.ConvDesc.k_valid.offset, L.TrDecoder.out_w.offset, L.LstmDecoder.out_k.offset] == \
        [int(v) for v in offs.split()]


def test_argument_errors_need_no_gpu():
    import image_captioning_amd as amd
    L = amd._lib
    assert L.lib.mic_version() >= 1
    d = L.ConvDesc()
    assert L.lib.mic_conv2d(C.byref(d), None) == -1
    assert b"null pointer" in L.lib.mic_last_error()
    assert L.lib.mic_logsoftmax_topk(None, 0, 0, 0, 3, 1.0, 0, None, None, None, None) == -1
    assert L.lib.mic_beam_select(None, None, None, None, None, None, None, 1, 3, 3, 0, 8, 2, 0, None) == -1
    assert L.lib.mic_pack_image(None, None, 0, 1, 8, 8, 0, 0, 0, 0, None) == -1
    assert L.lib.mic_linear_res_ln_supported(512) == 1 and L.lib.mic_linear_res_ln_supported(300) == 0
    assert L.lib.mic_resnet50_workspace_bytes(None, 1, 224, 224) == -1
    net = L.Resnet50()
    net.dtype = L.BF16
    b256 = L.lib.mic_resnet50_workspace_bytes(C.byref(net), 256, 224, 224)
    # packed image + 3 x (256*112*112*64) + 2 x (256*56*56*128) bf16 + the split-K scratch of the late layers
    assert 1_700_000_000 < b256 - L.lib.mic_conv2d_scratch_bytes() < 1_800_000_000
    sw = L.Swin()
    sw.dtype, sw.embed_dim = L.BF16, 96
    for i, (dp, hd) in enumerate(zip((2, 2, 18, 2), (3, 6, 12, 24))):
        sw.depths[i], sw.heads[i] = dp, hd
    per_token = 96 * (4 + 2 + 6 + 2 + 4 + 8)  # x, xt, qkv, o, u, hidden bytes per stage-0 token
    assert L.lib.mic_swin_workspace_bytes(C.byref(sw), 8, 224, 224) == 8 * 56 * 56 * per_token
    sw.heads[2] = 8
    assert L.lib.mic_swin_workspace_bytes(C.byref(sw), 8, 224, 224) == -1  # 384 channels need 12 heads of 32
    assert L.lib.mic_window_attn(None, None, None, None, None, 0, 1, 8, 8, 32, 1, 0, None) == -1
    assert L.lib.mic_patchify4(None, None, 0, 1, 8, 8, None) == -1
    # the graph-replayed LSTM search drivers validate before they touch the device
    assert L.lib.mic_lstm_beam_steps(None, None, None, None, None, None, None, None, 1, 3, 0, 1, 8, 2, 1, None) == -1
    dec, wss, st = L.LstmDecoder(), (L.LstmWorkspace * 2)(), (L.BeamState * 2)()
    wss[0].N = wss[1].N = 6
    wss[0].logits = wss[1].logits = 16
    assert L.lib.mic_lstm_beam_steps(C.byref(dec), wss, st, None, None, None, None, None, 2, 3, 0, 9, 8, 2, 1, None) == -1
    assert b"do not fit t_cap" in L.lib.mic_last_error()
    assert L.lib.mic_lstm_beam_steps(C.byref(dec), wss, st, 16, 16, 16, 16, 16, 2, 4, 0, 1, 8, 2, 1, None) == -1
    assert b"B*K" in L.lib.mic_last_error()
    assert L.lib.mic_lstm_sample_steps(C.byref(dec), wss, None, None, None, 1.0, None, None, None, 0, 1, 8, 0, 2, 1, None) == -1
    with pytest.raises(L.MicError, match="argument error"):
        L.check(-1, "probe")


def test_decode_roofline_bytes():
    """bench.py's search-phase byte count, term by term, at the headline configuration: the algorithmic figure is that of
    the unfolded layout (what the reference's mathematics needs), the folded layout the kernels use is reported beside it."""
    import bench
    r = bench.decode_roofline(8.0, 256, 3, 30, 49)
    N, H, V, Lr, e, heads = 768, 512, 3500, 3, 2, 2
    weights = (Lr * 12 * H * H + V * H) * e
    cross = Lr * 2 * 256 * 49 * H * e
    cache = Lr * 2 * N * H * e * sum(range(1, 31))
    logits = 2 * N * V * 4
    total = 30 * (weights + cross + logits) + cache + 256 * 49 * H * e + cross
    assert r["algorithmic_bytes"] == total
    assert r["unit"] == "TB/s" and abs(r["achieved"] - total / 8e-3 / 1e12) < 1e-3 and r["frac"] == round(r["achieved"] / 8.0, 4)
    w_f = (Lr * (4 + 2 * heads) * H * H + V * H) * e
    cache_f = Lr * (1 + heads) * N * H * e * sum(range(1, 31))
    assert r["traffic_model"] == 30 * (w_f + heads * cross + logits) + cache_f + 256 * 49 * H * e + cross + heads * cross
    assert r["traffic_model"] > r["algorithmic_bytes"]


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The host wrapper has no fallback: without the .so the import raises."""
    import importlib.util
    import image_captioning_amd as amd
    src = os.path.join(os.path.dirname(amd._lib.__file__), "_lib.py")
    dst = tmp_path / "_lib_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        spec.loader.exec_module(mod)
