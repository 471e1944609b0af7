/*
 * libmic_hip — C-ABI of the MI355X (gfx950) caption-generation kernels.
 *
 * Drop-in boundary for the generate path of MicaKil/image-captioning: everything
 * `model.generate(...)` computes between "images on the device" and "token ids +
 * scores (+ attention) on the device".  The reference has no native code; each entry
 * point below names the Python it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless the comment says "host";
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *    and nothing is allocated;
 *  - return value: 0 = ok, MIC_ERR_ARG (-1) = rejected arguments, >0 = hipError_t;
 *    `mic_last_error()` returns a thread-local message for the last non-zero return;
 *  - dtype codes select the storage/MFMA type of weights and GEMM inputs: MIC_F32 runs the
 *    f32-input MFMA (exact fp32 products), MIC_BF16 the bf16 MFMA with fp32 accumulation.
 *    Residual streams, LayerNorm, softmax, logits, scores stay fp32; beam scores are fp64;
 *  - activations are NHWC ("pixel-major"): an image-token matrix (B, S, C) is the same bytes.
 */
#ifndef MIC_HIP_H
#define MIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIC_F32 0
#define MIC_BF16 1
#define MIC_ERR_ARG (-1)

/* activation codes of the GEMM epilogues (the `relu` field / argument) */
#define MIC_ACT_NONE 0
#define MIC_ACT_RELU 1
#define MIC_ACT_GELU 2      /* x/2 * (1 + erf(x / sqrt 2)); mic_conv2d only */

#define MIC_MAX_BEAM 8      /* beam width accepted by the search kernels */
#define MIC_FOLD_MAX_HEADS 2 /* head count up to which the cross-attention projections are folded into K/V */
#define MIC_MAX_LAYERS 8    /* decoder depth accepted by the step drivers (sweeper/config.py:80 tops out at 6) */

const char* mic_last_error(void);
int mic_version(void);

/* ------------------------------------------------------------------------------------------
 * Encoder building blocks (models/encoders/ -> torchvision resnet50 children)
 * ---------------------------------------------------------------------------------------- */

/* Implicit-GEMM convolution / linear layer on the matrix cores:
 *   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + res[m, n] )
 * m enumerates output pixels (b, oh, ow); k enumerates (tap, c) with `run` contiguous input
 * elements per tap, tap = kh * kw_taps + kw read at input pixel (oh*stride - pad_h + kh,
 * ow*stride - pad_w + kw) (zero outside [0,IH)x[0,IW)).  A linear layer is the 1x1 case with
 * OH = OW = IH = IW = 1 and in_img_stride = the row stride of the input matrix.
 * Replaces nn.Conv2d + eval BatchNorm2d (folded into W/bias by the host) + ReLU + the bottleneck
 * residual add (torchvision Bottleneck.forward, reached from models/encoders/transformer.py:46),
 * the 1x1 projection (models/encoders/transformer.py:33-37,47) and every nn.Linear on the path
 * (models/transformer.py:83,136,175,293; models/intermediate.py:55-56,66; models/basic.py:49).
 * Requirements: K = taps*run, run % (64/sizeof(T)) == 0, W has n_pad >= ceil(N/128)*128 rows of K
 * elements (rows >= N zero), all base pointers 16-byte aligned. */
typedef struct mic_conv_desc {
    const void* in;       /* T, NHWC */
    const void* w;        /* T, [n_pad][K] */
    const float* bias;    /* fp32 [N] or NULL */
    const void* res;      /* residual [M][ld_res] of res_dtype, or NULL */
    void* out;            /* [M][ldc] of out_dtype */
    int32_t dtype;        /* MIC_F32 / MIC_BF16: type of in and w */
    int32_t out_dtype, res_dtype;
    int32_t M, N, K;
    int32_t OH, OW, IH, IW;
    int64_t in_img_stride, in_row_stride; /* elements */
    int32_t in_pix_stride;
    int32_t kw_taps, run;
    int32_t stride, pad_h, pad_w;
    int32_t ldc, ld_res;
    int32_t relu;         /* MIC_ACT_* */
    int32_t k_valid;      /* k entries that are not structural zero padding (0 = K); only used for FLOP accounting */
    /* Optional second K segment (NULL = none): the sum continues over k in [K, K+K2) with
     * A[m, k] = in2[b, oh*stride2, ow*stride2, k-K] (a 1x1 read, no padding) and W has K+K2 columns.
     * Fuses the stride-s 1x1 shortcut convolution of a bottleneck's first block into its last
     * convolution (torchvision Bottleneck.forward: out = conv3(...) + downsample(x)), so the shortcut
     * never round-trips through memory.  K and K2 must be multiples of 64/sizeof(T). */
    const void* in2;
    int64_t in2_img_stride, in2_row_stride;
    int32_t in2_pix_stride, stride2, K2;
    int32_t w_rows;       /* rows the weight matrix really has (0 = not stated): when stated, a matrix shorter than
                             ceil(N/128)*128 rows is rejected instead of being read past its end */
    /* Optional LayerNorm epilogue (ln_gamma NULL = none), for N the epilogue can hold in one tile (mic_conv2d_ln_supported):
     *     out[m, :] = res[m, :] + LayerNorm(sum_k A W + bias)[m, :] * ln_gamma + ln_beta      (eps 1e-5, fp32)
     * and ln_out[m, :] = the same values in `dtype` (NULL = not wanted).  out must be fp32, res fp32 or NULL and may be
     * the same buffer as out, relu must be MIC_ACT_NONE.  This is the residual-post-norm of a Swin-V2 block,
     * x = x + norm(branch(x)) (torchvision SwinTransformerBlockV2.forward, reached from models/encoders/swin.py:61-71),
     * and with res NULL the LayerNorm after the patch embedding and after each patch merge.  The result equals
     * mic_conv2d with an fp32 output followed by mic_layernorm / mic_layernorm_add bit for bit. */
    const float* ln_gamma;
    const float* ln_beta;
    void* ln_out;
    int32_t ld_ln_out;
    /* Optional scratch (NULL = none) that lets a layer with fewer 256 x 256 output tiles than the GPU has compute units
     * split K over several workgroups per tile (bf16, tile variant 18): 4 KiB of tickets, which must be zero before the
     * first launch and are left zero by every launch, then one fp32 partial tile of 256 KiB per (tile, slice);
     * mic_conv2d_scratch_bytes() is the size that serves every layer (a split never has more than 256 slices).  Launches that
     * share a scratch must be ordered on one stream.  Results of a split layer agree with the unsplit one to fp32 rounding of
     * the regrouped sum (the bf16 outputs can differ in the last bit).  No reference counterpart. */
    void* scratch;
    int64_t scratch_bytes;
} mic_conv_desc;
int mic_conv2d(const mic_conv_desc* d, void* stream);
int mic_conv2d_ln_supported(int N);
int64_t mic_conv2d_scratch_bytes(void);
/* The same layer on a named tile variant instead of the one the shape heuristic picks (0: 128x128, 1: 128x128 with
 * 128-byte stages, 4: 64x64, 8: 128x64, 11: 128x256, 16: 256x256 filled directly into LDS, 18: 256x256 in eight phases (bf16), ...; the list is in
 * csrc/igemm.hip).  Every variant computes the same result for every shape; this entry exists for tuning sweeps and
 * for per-variant parity tests.  No reference counterpart. */
int mic_conv2d_variant(const mic_conv_desc* d, int variant, void* stream);

/* The ResNet stem in one launch (bf16): 7x7 stride-2 convolution (BatchNorm folded) + ReLU + 3x3 stride-2 max-pool with
 * padding 1 (torchvision resnet50 conv1, bn1, relu, maxpool, reached from models/encoders/transformer.py:46).  `packed` is
 * the zero-bordered 4-channel image mic_pack_image writes ([B][Hp][Wp][4], border 3 on the top and left, Wp even), w the
 * stem's weight matrix [>= 64 rows][7*32] with k = kernel row * 32 + pixel * 4 + channel, out [B][H2][W2][64] with
 * H2 = (H1 - 1) / 2 + 1 for the H1 x W1 convolution output that never reaches memory.  Equals mic_conv2d followed by
 * mic_maxpool3x3s2 bit for bit. */
int mic_stem_pool_supported(int dtype);
int mic_stem_pool(const void* packed, const void* w, const float* bias, void* out, int dtype, int B, int Hp, int Wp, int H1, int W1,
                  int w_rows, void* stream);
/* The same launch followed, on the pooled tile while it is still on chip, by the first convolution of the bottleneck
 * that comes next (1x1, 64 -> 64, BatchNorm folded, ReLU; torchvision layer1[0].conv1/bn1/relu): t1 [B][H2][W2][64] =
 * relu(out * w1^T + b1) with w1 [>= 64 rows][64].  Equals mic_stem_pool followed by mic_conv2d bit for bit. */
int mic_stem_pool_conv1(const void* packed, const void* w, const float* bias, void* out, const void* w1, const float* b1, void* t1,
                        int dtype, int B, int Hp, int Wp, int H1, int W1, int w_rows, int w1_rows, void* stream);

/* Tail of one ResNet bottleneck and head of the next in ONE launch (bf16), for identity blocks (stride 1, no
 * shortcut convolution):
 *     t2  = relu(conv3x3_pad1(t1) + b2)        C -> C channels, kept on chip
 *     x'  = relu(conv1x1(t2) + b3 + x)         C -> 4C, written to xout (the next block's residual)
 *     t1' = relu(conv1x1(x') + b1n)            4C -> C2, written to t1n (the next block's 3x3 input); C2 == 0: skipped
 * i.e. conv2/bn2/relu, conv3/bn3, the residual add and ReLU of torchvision Bottleneck.forward followed by
 * conv1/bn1/relu of the NEXT Bottleneck (reached from models/encoders/transformer.py:46), BatchNorm folded into
 * the weights by the host.  Same arithmetic and the same bf16 rounding points as three mic_conv2d launches; what
 * changes is the memory traffic: t2 never leaves the chip and x' is written once instead of written and re-read.
 * Activations NHWC; w2 [>=128 rows][9C] with k = (kh*3+kw)*C + c; w3 [4C][C]; w1n [>=128 rows][4C]; biases fp32.
 * Supported (mic_bottleneck_fused_supported): bf16, (C, C2) in {(64,0), (64,64), (64,128), (128,0), (128,128)}. */
typedef struct mic_fused_block_desc {
    const void* t1;       /* [B][H][W][C] */
    const void* w2; const float* b2;
    const void* w3; const float* b3;
    const void* res;      /* [B*H*W][4C] */
    void* xout;           /* [B*H*W][4C] */
    const void* w1n; const float* b1n;   /* NULL when C2 == 0 */
    void* t1n;            /* [B*H*W][C2] */
    int32_t dtype, B, H, W, C, C2;
    int32_t w2_rows, w3_rows, w1n_rows;  /* rows the weight matrices really have */
    /* First block of a stage with a stride-1 shortcut convolution (out = conv3(...) + downsample(x), torchvision
     * Bottleneck.forward): CS > 0 channels of xs [B*H*W][CS] are a second K segment of the expansion, w3 is then
     * [4C][C + CS] = [W3 | W_downsample] with the summed bias in b3, and res must be NULL
     * (mic_bottleneck_fused_shortcut_supported: bf16, C = C2 = CS = 64, the first block of ResNet-50's first stage). */
    const void* xs;
    int32_t CS;
} mic_fused_block_desc;
int mic_bottleneck_fused_supported(int dtype, int C, int C2);
int mic_bottleneck_fused_shortcut_supported(int dtype, int C, int C2, int CS);
int mic_bottleneck_fused(const mic_fused_block_desc* d, void* stream);

/* Per-launch timing of the GEMM/conv kernel for roofline reports.  Between begin and end every
 * mic_conv2d launch is bracketed by two hipEvents on its stream (no synchronisation).  end waits for
 * the last event and returns, per tile variant v (0: 128x128, 1: 64x64) and dtype d (index 2*v+d),
 * the number of launches, their summed duration in ms and their summed algorithmic FLOPs
 * 2*M*N*k_valid.  No reference counterpart (the reference has no timing hooks); used by bench.py. */
int mic_profile_begin(int max_launches);
int mic_profile_end(int32_t count[4], double ms[4], double flops[4]);

/* (B,3,H,W) fp32 planar image -> zero-bordered 4-channel pixel-major image
 * out[B][H+pad_t+pad_b][W+pad_l+pad_r][4] of `dtype` (channel 3 = 0).  Input staging for the
 * 7x7/2 stem, which then reads each kernel row as one contiguous 32-element run.
 * Replaces the tensor handed to resnet conv1 (models/encoders/transformer.py:46). */
int mic_pack_image(const float* nchw, void* out, int dtype, int B, int H, int W,
                   int pad_t, int pad_b, int pad_l, int pad_r, void* stream);

/* Input transform of the reference on the device (runner/config.py:7-12: v2.ToImage, v2.Resize((256,256)),
 * v2.ToDtype(float32, scale=True), v2.Normalize): src uint8 [B][H][W][3] (decoded RGB, interleaved) ->
 * dst fp32 [B][3][OH][OW] = (resize(src) / 255 - mean) / std with the antialiased bilinear (triangle)
 * filter of interpolate(mode="bilinear", antialias=True), which is what v2.Resize applies to tensors.
 * round_u8 != 0 rounds the resized value to an integer level first, as the reference's uint8 intermediate
 * does.  mean / std are HOST arrays of 3.  Downscale factors up to 64 per axis. */
int mic_resize_normalize(const uint8_t* src, float* dst, int B, int H, int W, int OH, int OW, const float mean[3],
                         const float std[3], int round_u8, void* stream);

/* 3x3 stride-2 pad-1 max pooling, NHWC (torchvision resnet50 child 3, nn.MaxPool2d(3,2,1)). */
int mic_maxpool3x3s2(const void* in, void* out, int dtype, int B, int H, int W, int C, void* stream);

/* Global average pooling NHWC (B,HW,C) -> (B,C) (torchvision resnet50 child 8,
 * AdaptiveAvgPool2d(1), kept by models/encoders/intermediate.py:27-28 and basic.py:25-26).
 * ld_out is the row stride of out in elements. */
int mic_global_avgpool(const void* in, void* out, int dtype, int B, int HW, int C, int ld_out, void* stream);

/* (B,3,H,W) fp32 planar image -> non-overlapping 4x4 patches [B][H/4][W/4][64] of `dtype`,
 * element (kh*4 + kw)*4 + c with a zero 4th channel: the im2col of the 4x4/4 patch-embedding
 * convolution (torchvision swin features[0][0], models/encoders/swin.py:26,68), which then is a
 * plain K=64 GEMM through mic_conv2d. */
int mic_patchify4(const float* images, void* out, int dtype, int B, int H, int W, void* stream);

/* Shifted-window cosine attention of Swin-V2 for every (image, 8x8 window, head):
 *   o = softmax( normalize(q) normalize(k)^T * logit_scale[h] + bias_tab[h] + shift_mask ) v
 * (torchvision shifted_window_attention with logit_scale, between the qkv and proj linears;
 * models/encoders/swin.py:68).  qkv [B*H*W][3C] of `dtype` is the packed projection of the
 * UNPADDED, UNSHIFTED token map; zero padding of H, W to multiples of 8, the cyclic shift by
 * `shift` (dropped on an axis one window covers), window partition and their inverses are index
 * arithmetic inside the kernel.  Padded slots take qkv_bias (fp32 [3C], key third zero) as their
 * projection.  bias_tab fp32 [heads][64][64] = 16*sigmoid(cpb_mlp(table))[index]; logit_scale fp32
 * [heads] = exp(min(param, log 100)); both precomputed by the host at load time.  C = heads*32.
 * o [B*H*W][C] of `dtype` (input of the proj GEMM). */
int mic_window_attn(const void* qkv, const float* qkv_bias, const float* bias_tab, const float* logit_scale, void* o,
                    int dtype, int B, int H, int W, int C, int heads, int shift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder step building blocks
 * ---------------------------------------------------------------------------------------- */

/* x[n,:] = tok_emb[ids[n*ld_ids + pos]] + pos_emb[pos]; writes fp32 x and (if xt != NULL) a copy in
 * `dtype` with row stride ld_xt.  pos_emb == NULL gives a plain embedding gather (LSTM decoders).
 * Replaces SeqEmbedding.forward (models/transformer.py:44-50) for the newest position and
 * nn.Embedding in models/basic.py:43, models/intermediate.py:59. */
int mic_embed_pos(const int32_t* ids, int ld_ids, int pos, const float* tok_emb, const float* pos_emb,
                  float* x, void* xt, int ld_xt, int dtype, int N, int H, int vocab, void* stream);

/* Row LayerNorm (eps 1e-5): y = LN(x) * gamma + beta; writes fp32 y (may alias x) and optional
 * `dtype` copy yt (row stride ld_yt).  nn.LayerNorm at models/transformer.py:86,140,177 and
 * models/intermediate.py:64. */
int mic_layernorm(const float* x, const float* gamma, const float* beta, float* y, void* yt, int ld_yt,
                  int dtype, int N, int H, void* stream);

/* Residual-post-norm update y = res + LN(u) * gamma + beta (eps 1e-5): the block form of Swin-V2,
 * x = x + norm(attn(x)) / x = x + norm(mlp(x)) (torchvision SwinTransformerBlockV2.forward, reached
 * from models/encoders/swin.py:68).  y may alias res; u must not alias y.  Optional `dtype` copy yt. */
int mic_layernorm_add(const float* u, const float* gamma, const float* beta, const float* res, float* y, void* yt,
                      int ld_yt, int dtype, int N, int H, void* stream);

/* Fused tail of a post-LN transformer sub-layer in one launch:
 *   y = LayerNorm(x W^T + bias + res) * gamma + beta,   eps 1e-5
 * x [M][ld_x] in `dtype`; W is the [H][K] weight in FRAGMENT ORDER as written by
 * mic_pack_weight_frag; res fp32 [M][ld_res] (may alias y: each workgroup reads
 * the rows it owns before writing them); writes fp32 y [M][H] and/or a `dtype` copy yt [M][ld_yt].
 * Replaces out_proj / ff.seq.2 + the residual add + nn.LayerNorm (models/transformer.py:83-86,
 * :136-140, :175-177) for the newest position.  Available for H in {64,128,256,512,1024}
 * (mic_linear_res_ln_supported); other sizes use mic_conv2d + mic_layernorm. */
int mic_linear_res_ln_supported(int H);
/* [N][K] row-major weight -> [N/16][K/e][64][16 bytes] (e = elements per 64 bytes): the 1 KiB a
 * wave loads as one MFMA operand (lane = (row & 15) + 16 * (16-byte chunk of the 64-byte k-tile))
 * becomes contiguous.  Load-time layout change of nn.Linear weights; no reference counterpart. */
int mic_pack_weight_frag(const void* w, void* out, int dtype, int N, int K, void* stream);
int mic_linear_res_ln(const void* x, int ld_x, const void* w, const float* bias, const float* res, int ld_res,
                      const float* gamma, const float* beta, float* y, void* yt, int ld_yt, int dtype, int M, int K, int H,
                      void* stream);

/* out[M][ldc] = act(x W^T + bias) for a few hundred rows: workgroups of 16/32 rows x 128 columns
 * stream fragment-ordered weights (mic_pack_weight_frag of the [N padded to 128][K] matrix, padding
 * rows zero) straight into MFMA operands.  Same math as mic_conv2d's linear case; used by the step
 * drivers for the in-projections, ff.seq.0 and the vocabulary projection
 * (models/transformer.py:83,136,175,293) when the row count is small. */
int mic_linear_skinny(const void* x, int ld_x, const void* wf, const float* bias, void* out, int out_dtype, int ldc,
                      int dtype, int M, int N, int K, int relu, void* stream);

/* General form of the skinny GEMM:  out = act(X W^T + bias + R).  X and/or R may be a LayerNorm VIEW
 * of a raw fp32 matrix, LN(raw) * gamma + beta (eps 1e-5), evaluated on the fly from per-row partial
 * statistics: stats[m][p] = (mean, centred sum of squares) of raw[m][128p .. 128p+127].  A launch
 * writes those statistics for its own fp32 output when stats_out != NULL ([M][ceil(N/128)][2]).
 * Chaining launches this way removes every nn.LayerNorm launch of a post-LN decoder layer
 * (models/transformer.py:86,140,177): the producer stores the pre-norm sum and its statistics, the
 * consumers (next projection, next residual add) normalise while they load.  out may alias
 * res_ln.raw (each element is read and written by the same thread). */
typedef struct mic_ln_ref {
    const float* raw;     /* [M][ld] pre-LayerNorm values, NULL = not a view */
    const float* stats;   /* [M][parts][2] */
    const float *gamma, *beta;
    int32_t ld, parts, width;  /* parts == ceil(width / 128) */
} mic_ln_ref;

typedef struct mic_skinny_desc {
    const void* x;        /* [M][ld_x] in dtype (used when x_ln.raw == NULL) */
    mic_ln_ref x_ln;      /* x_ln.width == K */
    const void* wf;       /* fragment-ordered [N padded to 128][K] weight */
    const float* bias;
    const float* res;     /* plain fp32 residual [M][ld_res] or NULL */
    mic_ln_ref res_ln;    /* res_ln.width == N */
    void* out;            /* [M][ldc] of out_dtype */
    float* stats_out;     /* NULL, or [M][ceil(N/128)][2] (fp32 output, no ReLU) */
    int32_t dtype, out_dtype, ld_x, ld_res, ldc, M, N, K, relu;
} mic_skinny_desc;
int mic_linear_skinny_ex(const mic_skinny_desc* d, void* stream);

/* The gate GEMM of an LSTM layer with the cell in its epilogue (models/intermediate.py:59-63, models/basic.py:45-55:
 * nn.LSTM's i, f, g, o from [x | h] W^T + b, then c' = f c + i g, h' = o tanh(c')).  d as for mic_linear_skinny_ex with
 * N = 4 Hd and the weight rows / bias in UNIT-MAJOR order (row 4u + g = gate g of hidden unit u, g = 0..3 for input,
 * forget, cell, output), so that a lane's four accumulators are the four gates of one unit; no activation, residual or
 * statistics; d->out is not written.  c [M][Hd] fp32 is updated in place, h' goes to h [M][Hd] fp32 and, in d->dtype, to
 * ht0 [M][ld0] and ht1 [M][ld1] (either may be NULL).  Every workgroup of the launch reads the rows of d->x, so ht0 / ht1
 * must not overlap them (checked): the recurrent half of [x | h] is refreshed by whoever moves the state to the next step. */
int mic_lstm_gates_cell(const mic_skinny_desc* d, float* c, float* h, void* ht0, int ld0, void* ht1, int ld1, int Hd, void* stream);

/* One new query per row against that row's cached keys/values, all heads
 * (the attention inside nn.MultiheadAttention, torch/nn/functional.py:6578-6606, for the last
 * text position only; the rest of the prefix is what the cache holds).
 *  self form  (models/transformer.py:83): qkv fp32 [N][3H] = packed in-proj output for position t.
 *      k,v of position t are appended to kcache/vcache [N][t_cap][H] (`dtype`), and positions
 *      p < t are read from row anc[n*t_cap + p] (beam ancestry; anc == NULL reads row n).
 *  cross form (models/transformer.py:136): q fp32 [N][ldq]; keys/values of row n are
 *      kv[(n / rows_per_image)][S][2H] (`dtype`, k then v).  If attn_out != NULL the head-averaged
 *      weights (N,S) are written as attn_out = (accumulate ? attn_out : 0) + attn_scale * w.
 * Output o [N][ld_o] in `dtype` (input of the out-projection GEMM). */
int mic_self_attn_step(const float* qkv, void* kcache, void* vcache, const int32_t* anc, int t, int t_cap,
                       void* o, int ld_o, int dtype, int N, int H, int heads, void* stream);
int mic_cross_attn_step(const float* q, int ldq, const void* kv, int S, int rows_per_image,
                        void* o, int ld_o, float* attn_out, int ld_attn, float attn_scale, int accumulate,
                        int dtype, int N, int H, int heads, void* stream);

/* The whole cross-attention sub-layer of a post-LN decoder layer in ONE launch (models/transformer.py:
 * 130-140: mha(query=x, key=value=image tokens) -> + x -> LayerNorm, the norm left to the consumer):
 *     y = x + sum_h softmax_s( x . K'_h[s] + c_h[s] ) V'_h[s] + out_bias,     x = LN(x_ln->raw)
 * with the query projection folded into per-image keys and the output projection into per-image values,
 *     K'_h[s] = W_q,h^T k_h[s] / sqrt(d),  c_h[s] = b_q,h . k_h[s] / sqrt(d),  V'_h[s] = W_o[:, h-th slice] v_h[s],
 * (kf, vf: [B][S][heads][H] of `dtype`; c: fp32 [B][S][heads]) which mic_tr_cross_kv prepares once per image
 * when the decoder carries the folded weights.  Same mathematics as in-projection + attention + out-projection,
 * re-associated; it removes two GEMM launches per layer per step.  Writes fp32 y [N][H] (may alias
 * x_ln->raw) and, when stats_out != NULL, its row statistics per 128 columns (see mic_ln_ref).
 * attn_out as in mic_cross_attn_step.  Supported (mic_cross_attn_folded_supported): H == 1024 bytes / sizeof
 * element per row (512 bf16, 256 fp32), 1 or 2 heads, S <= 64, up to MIC_MAX_BEAM rows per image. */
int mic_cross_attn_folded_supported(int dtype, int H, int heads, int S, int rows_per_image);
int mic_cross_attn_folded(const mic_ln_ref* x_ln, const void* kf, const void* vf, const float* c, const float* out_bias, int S,
                          int rows_per_image, float* y, float* stats_out, float* attn_out, int ld_attn, float attn_scale,
                          int accumulate, int dtype, int N, int H, int heads, void* stream);
/* c[row][h] = q_bias_scaled[h-th slice] . k[row][h-th slice] for `rows` key rows of kv [rows][2H] (k then v). */
int mic_cross_fold_bias(const void* kv, const float* q_bias_scaled, float* c, int dtype, int rows, int H, int heads, void* stream);
/* The same constant from the image tokens: c[row][h] = feats[row] . u[h] + c0[h] (u: [heads][H] fp32), for decoders that carry
 * the combined matrices of mic_tr_layer (cross_cu, cross_c0).  No reference counterpart beyond the bias term of
 * nn.MultiheadAttention's query projection (models/transformer.py:136). */
int mic_cross_fold_c(const void* feats, const float* u, const float* c0, float* c, int dtype, int rows, int H, int heads, void* stream);

/* Both attention sub-layers of a post-LN decoder layer for one new position in ONE launch, one workgroup per image
 * (models/transformer.py:81-86 CausalSelfAttention on the newest position, :100 its residual + LayerNorm, :136-140
 * CrossAttention, its residual; the second LayerNorm is left to the consumer):
 *     y1 = x + sum_h softmax_p( q_h . k_h[p] / sqrt(d) ) V'_h[p] + self_b,   p over the row's cached positions 0..t
 *     u  = LayerNorm(y1) * ln1_g + ln1_b
 *     y  = u + sum_h softmax_s( u . K'_h[s] + c_h[s] ) V'x_h[s] + bo          (cross side: as mic_cross_attn_folded)
 * with the self-attention output projection folded into the cached values: the caller's packed in-projection yields
 * qkvf [N][(2+heads)H] fp32 = q | k | v'_0 | .. | v'_{heads-1}, v'_h = (W_o[:, h-th slice] W_v[h-th slice, :]) x, and
 * self_b = b_o + sum_h W_o[:, h-th slice] b_v[h-th slice] (softmax weights sum to one).  Position t of every row is
 * appended to kcache [N][t_cap][H] and vfcache [N][t_cap][heads][H] (`dtype`); earlier positions are read through the
 * ancestry table anc [N][t_cap] (NULL: own row), whose entries must name rows of the same image.  x is either plain
 * fp32 rows (x) or a LayerNorm view (x_ln; ld == width == H), exactly one of the two.  Writes fp32 y [N][H] (may
 * alias x_ln.raw) and, when stats_out != NULL, its row statistics per 128 columns (see mic_ln_ref).  attn_out as in
 * mic_cross_attn_step.  Same mathematics as the three launches it replaces (mic_self_attn_step, the output
 * projection, mic_cross_attn_folded), re-associated.  Supported (mic_attn_fused_supported): H == 1024 bytes / sizeof
 * element per row (512 bf16, 256 fp32), 1 or 2 heads, S <= 64, t_cap <= 48 (six cached positions per wave above 32, four
 * up to it), up to MIC_MAX_BEAM rows per image. */
typedef struct mic_attn_fused_desc {
    const float* qkvf;
    void *kcache, *vfcache;
    const int32_t* anc;
    const float* x;
    mic_ln_ref x_ln;
    const float *self_b, *ln1_g, *ln1_b;
    const void *kf, *vf;      /* folded cross keys / values [N/rows_per_image][S][heads][H] */
    const float *c, *bo;
    float *y, *stats_out, *attn_out;
    int32_t ld_attn, accumulate;
    float attn_scale;
    int32_t dtype, N, H, heads, S, rows_per_image, t, t_cap;
} mic_attn_fused_desc;
int mic_attn_fused_supported(int dtype, int H, int heads, int S, int rows_per_image, int t_cap);
int mic_attn_fused(const mic_attn_fused_desc* d, void* stream);

/* LSTM cell pointwise part, gate order i,f,g,o (nn.LSTM, models/basic.py:48,
 * models/intermediate.py:63): c' = s(f)*c + s(i)*tanh(g); h' = s(o)*tanh(c').  gates fp32
 * [N][4*Hd]; c, h fp32 [N][Hd] updated in place; h' is also written in `dtype` to ht0 (row stride
 * ld0) and ht1 (row stride ld1) when non-NULL: the recurrent half of this layer's next GEMM input
 * and the input half of the layer above. */
int mic_lstm_cell(const float* gates, float* c, float* h, void* ht0, int ld0, void* ht1, int ld1,
                  int dtype, int N, int Hd, void* stream);

/* dst[r,:] = src[idx[r],:] for `rows` rows of `row_bytes` bytes (multiple of 4):
 * beam reorder of per-beam recurrent state (the list slicing at models/image_captioner.py:148-149). */
int mic_gather_rows(const void* src, void* dst, const int32_t* idx, int rows, int row_bytes, void* stream);
/* The same for up to MIC_GATHER_MAX buffers in one launch (every state tensor of every layer). */
int mic_gather_rows_multi(const void* const* src, void* const* dst, const int32_t* row_bytes, int n_bufs, const int32_t* idx,
                          int rows, void* stream);

/* mic_lstm_cell of the top layer followed by the LayerNorm of the intermediate decoder (models/intermediate.py:63,65) in one
 * launch: h' also goes to ht0 (the layer's own recurrent input), hn / hnt receive LayerNorm(h') in fp32 / `dtype`
 * (either may be NULL).  Hd % 4 == 0, Hd <= 1024. */
int mic_lstm_cell_ln_supported(int Hd);
int mic_lstm_cell_ln(const float* gates, float* c, float* h, void* ht0, int ld0, const float* ln_g, const float* ln_b, float* hn,
                     void* hnt, int ld_hnt, int dtype, int N, int Hd, void* stream);

/* ------------------------------------------------------------------------------------------
 * Search kernels
 * ---------------------------------------------------------------------------------------- */

/* Per row: lp = log_softmax(logits * inv_temp) and its K largest entries, largest first, ties to
 * the lower index.  rank_raw != 0 ranks by the raw logits instead of lp (argmax of logits,
 * models/image_captioner.py:92).  Replaces F.log_softmax + topk/argmax
 * (models/transformer.py:516,526,610,615; models/image_captioner.py:138-139).
 * out_val fp32 [N][K], out_idx int32 [N][K]; lse (optional) fp32 [N] = max + log(sum exp). */
int mic_logsoftmax_topk(const float* logits, int ld, int N, int V, int K, float inv_temp, int rank_raw,
                        float* out_val, int32_t* out_idx, float* lse, void* stream);

/* Inverse-CDF draw from softmax(logits*inv_temp) with caller-supplied uniforms u[N]: first index
 * whose running fp32 sum of probabilities exceeds u.  Writes the index and its log-probability.
 * Stands in for torch.multinomial (models/transformer.py:521-522, models/image_captioner.py:94-95),
 * whose random stream cannot be reproduced on the device. */
int mic_sample_multinomial(const float* logits, int ld, int N, int V, float inv_temp, const float* u,
                           float* out_val, int32_t* out_idx, void* stream);

/* Beam state of B images x K slots.  seq/anc are [B*K][t_cap] int32; score fp64; len int32. */
typedef struct mic_beam_state {
    double* score;
    int32_t* len;
    int32_t* seq;
    int32_t* anc;     /* may be NULL (LSTM path keeps no ancestry) */
} mic_beam_state;

/* Start of a search: both states get score 0, length 1, every sequence position the start token, ancestry 0 (when kept),
 * and the t_cap live-image counters are cleared (live_images may be NULL) - the list initialisation at
 * models/transformer.py:589-596 and models/image_captioner.py:117-121 for all images in one launch. */
int mic_beam_init(const mic_beam_state st[2], int32_t* live_images, int N, int t_cap, int sos_id, void* stream);

/* One beam-search selection step for every image (models/transformer.py:613-628 when
 * finished_in_place == 0, models/image_captioner.py:128-149 when 1): extend each live slot by its K
 * best tokens (top_val/top_idx [B*K][K]), pool with finished slots, stable-sort by
 * score / sqrt_tab[len] descending (fp64; sqrt_tab[i] = i ** 0.5 computed by the host), keep K.
 * Only the first n_live_slots slots of `cur` are considered (1 at the first step).  Writes `nxt`,
 * parent[B*K] (slot each survivor came from) and adds the number of images that still have a live
 * beam to *live_images (caller zeroes it). */
int mic_beam_select(const mic_beam_state* cur, const mic_beam_state* nxt, const float* top_val,
                    const int32_t* top_idx, int32_t* parent, int32_t* live_images,
                    const double* sqrt_tab, int B, int K, int n_live_slots, int t, int t_cap,
                    int eos_id, int finished_in_place, void* stream);

/* The tail of one transformer beam-search step in ONE launch per image: mic_logsoftmax_topk (inv_temp 1) over the
 * image's K rows of logits [B*K][ld], mic_beam_select (finished_in_place == 0) on the result, and, when tok_emb != NULL,
 * mic_embed_pos of the tokens just chosen for position t + 1 into x (fp32 [B*K][H]) / xt (`dtype`), i.e. the first
 * kernel of the next step (models/transformer.py:610-628, then :602 of the next iteration).  A finished beam's row is
 * embedded with the end token (its row is never read again).  Supported: K <= MIC_MAX_BEAM, V <= 4096. */
int mic_beam_tail_supported(int K, int V);
int mic_beam_tail(const float* logits, int ld, int V, const mic_beam_state* cur, const mic_beam_state* nxt, int32_t* parent,
                  int32_t* live_images, const double* sqrt_tab, int B, int K, int n_live_slots, int t, int t_cap, int eos_id,
                  const float* tok_emb, const float* pos_emb, float* x, void* xt, int dtype, int H, int vocab, void* stream);

/* The same launch in its general form: finished beams kept in place (finished_in_place = 1: the LSTM family's search,
 * models/image_captioner.py:107-159), an embedding without positional table (pos_emb NULL) written with its own row stride
 * (ld_xt elements of `dtype`: the input half of an LSTM layer's [x | h] matrix), and up to MIC_GATHER_MAX per-beam state
 * buffers moved with the survivors, dst[b*K + r, :row_bytes] = src[parent of (b, r), :row_bytes] with rows stride_bytes apart
 * (the list slicing at models/image_captioner.py:148-149).  With the gathers a beam-search step of the LSTM decoders ends in
 * this one launch instead of top-k + selection + one gather per state tensor + the next step's embedding. */
#define MIC_GATHER_MAX 16
typedef struct mic_beam_tail_desc {
    const float* logits;       /* [B*K][ld] fp32 */
    int32_t ld, V;
    const mic_beam_state* cur; /* state before the step */
    const mic_beam_state* nxt; /* state after it */
    int32_t* parent;           /* [B*K] out */
    int32_t* live_images;      /* one counter, incremented per image with a live beam, or NULL */
    const double* sqrt_tab;
    int32_t B, K, n_live_slots, t, t_cap, eos_id, finished_in_place;
    const float* tok_emb;      /* [vocab][H] fp32, NULL = no embedding */
    const float* pos_emb;      /* [positions][H] fp32 or NULL */
    float* x;                  /* [B*K][H] fp32 or NULL */
    void* xt;                  /* [B*K][ld_xt] dtype or NULL */
    int32_t ld_xt, dtype, H, vocab;
    int32_t n_gather;
    const void* gather_src[MIC_GATHER_MAX];
    void* gather_dst[MIC_GATHER_MAX];
    int32_t gather_row_bytes[MIC_GATHER_MAX], gather_stride_bytes[MIC_GATHER_MAX];
    int32_t row_group_threads;   /* threads that share one row of logits: 0 = 256 for up to four beams and 128 above; 128 may be
                                    asked for with three or four beams (the grouping of the one-launch decoder step's tail
                                    phase: the log-sum-exp is then summed in that order, so both give the same bits) */
} mic_beam_tail_desc;
int mic_beam_tail_ex(const mic_beam_tail_desc* d, void* stream);

/* Best beam per image (first maximum of score/sqrt(len), models/transformer.py:635-638):
 * tokens [B][t_cap] (-1 padded), lengths [B], score fp32 [B]; if attn_hist != NULL
 * ([B*K][t_cap][S], per-position layer-averaged weights) also attn_out [B][t_cap][S] with
 * attn_out[b][s] = mean over p <= s of hist[anc[best][p]][p] (models/transformer.py:606-611,620). */
int mic_beam_finalize(const mic_beam_state* st, const double* sqrt_tab, int B, int K, int t_cap,
                      int32_t* tokens, int32_t* lengths, float* score, const float* attn_hist, int S,
                      float* attn_out, void* stream);

/* Greedy / sampling bookkeeping for one step (models/transformer.py:530-543,
 * models/image_captioner.py:98-102): total += finished ? 0 : val; tok = finished ? pad : idx;
 * tokens[n*t_cap + pos] = tok; finished |= tok == eos; adds the number of unfinished rows to
 * *live_rows (caller zeroes it). */
int mic_sample_update(const float* val, const int32_t* idx, int32_t* tokens, int t_cap, int pos,
                      uint8_t* finished, float* total, int32_t* live_rows, int N, int pad_id, int eos_id,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-path drivers: the same kernels chained natively so the host issues one call per phase.
 * ---------------------------------------------------------------------------------------- */

/* ResNet-50 trunk (+ optional max-pool'd stem) as a list of fused conv layers.  The host fills one
 * mic_conv_layer per convolution (BN folded); the driver fixes up pointers and shapes per batch. */
typedef struct mic_conv_layer {
    const void* w;
    const float* bias;
    int32_t cin, cout, kh, kw, stride, pad, relu;
} mic_conv_layer;

/* conv3 and the shortcut convolution of a stage's first bottleneck as ONE GEMM over K = cin_main +
 * cin_skip: w = [conv3 | downsample] column-concatenated (BN folded), bias = their sum. */
typedef struct mic_fused_tail {
    const void* w;        /* NULL: run downsample and conv3 separately */
    const float* bias;
    int32_t cin_main, cin_skip, cout, skip_stride;
} mic_fused_tail;

typedef struct mic_resnet50 {
    int32_t dtype;
    mic_conv_layer stem;            /* 7x7/2 as 7 row-taps of 32 over the packed image */
    int32_t blocks_per_stage[4];
    mic_conv_layer conv[16][4];     /* per bottleneck: conv1, conv2, conv3, downsample (cout==0: none) */
    mic_fused_tail tail[4];         /* optional, per stage */
} mic_resnet50;

/* Bytes of workspace mic_resnet50_forward needs for a batch. */
int64_t mic_resnet50_workspace_bytes(const mic_resnet50* net, int B, int H, int W);
/* images fp32 (B,3,H,W) -> out [B][H/32*W/32][2048] in net->dtype.
 * torchvision resnet50 children[:-2] as used by models/encoders/transformer.py:24-28,46. */
int mic_resnet50_forward(const mic_resnet50* net, const float* images, int B, int H, int W,
                         void* workspace, int64_t workspace_bytes, void* out, void* stream);

/* Swin-V2 feature extractor (torchvision swin_v2_{t,s,b}.features as used by
 * models/encoders/swin.py:26-30,68): 4x4 patch embedding + LayerNorm, four stages of
 * residual-post-norm blocks with 8x8 (shifted) window cosine attention and a GELU MLP, 2x2 patch
 * merging (reduction + LayerNorm) between stages, no final norm. */
#define MIC_SWIN_MAX_BLOCKS 32
typedef struct mic_swin_block {
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;          /* dtype, rows padded to 128 */
    const float *qkv_b, *proj_b, *fc1_b, *fc2_b;         /* qkv_b: key third zero */
    const float *n1_g, *n1_b, *n2_g, *n2_b;
    const float* bias_tab;                               /* [heads][64][64] */
    const float* logit_scale;                            /* [heads] */
} mic_swin_block;

typedef struct mic_swin_merge {
    const void* w;        /* dtype [2C padded][4C], input blocks ordered (kh,kw) = (0,0),(0,1),(1,0),(1,1) */
    const float *g, *b;   /* LayerNorm over 2C */
} mic_swin_merge;

typedef struct mic_swin {
    int32_t dtype, embed_dim;
    int32_t depths[4], heads[4];                         /* heads[s] * 32 == embed_dim << s */
    const void* patch_w;                                 /* dtype [embed padded][64], k = (kh*4+kw)*4+c */
    const float *patch_b, *patch_g, *patch_be;           /* conv bias, LayerNorm weight / bias */
    mic_swin_block block[MIC_SWIN_MAX_BLOCKS];           /* stage-major */
    mic_swin_merge merge[3];
} mic_swin;

int64_t mic_swin_workspace_bytes(const mic_swin* net, int B, int H, int W);
/* images fp32 (B,3,H,W) -> out [B][ceil(H/32)*ceil(W/32)][8*embed_dim] in net->dtype (channels-last
 * tokens; H and W multiples of 4). */
int mic_swin_forward(const mic_swin* net, const float* images, int B, int H, int W, void* workspace,
                     int64_t workspace_bytes, void* out, void* stream);

typedef struct mic_tr_layer {
    const void *self_in_w, *self_out_w, *cross_q_w, *cross_kv_w, *cross_out_w, *ff1_w, *ff2_w; /* dtype, padded */
    const float *self_in_b, *self_out_b, *cross_q_b, *cross_kv_b, *cross_out_b, *ff1_b, *ff2_b;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    /* optional: self_out_w / cross_out_w / ff2_w, rows zero-padded to a multiple of 128, in fragment order
     * (mic_pack_weight_frag).  The step driver feeds them to mic_linear_skinny_ex (whose workgroups fetch
     * whole 128-column groups, hence the padding) or, above 2048 rows and when the hidden size has a
     * fused variant, to mic_linear_res_ln (which reads the first H rows only). */
    const void *self_out_wf, *cross_out_wf, *ff2_wf;
    /* optional: self_in_w / cross_q_w / ff1_w (rows padded to 128) in fragment order for
     * mic_linear_skinny; used when set and the row count is at most 2048. */
    const void *self_in_wf, *cross_q_wf, *ff1_wf;
    /* optional, for at most MIC_FOLD_MAX_HEADS heads: per-head matrices that fold the query projection into the
     * cross keys (cross_kf_w[h] = (W_q rows of head h)^T / sqrt(d), [H padded][d]) and the output projection into
     * the cross values (cross_vf_w[h] = W_o columns of head h, [H padded][d]), and the query bias / sqrt(d).
     * With them and the workspace buffers below the step driver runs mic_cross_attn_folded. */
    const void* cross_kf_w[2];
    const void* cross_vf_w[2];
    const float* cross_qb_scaled;
    /* Optional: the key / value projection and the fold above as ONE matrix each, applied to the image tokens directly:
     * cross_kfc_w = [W_q,h^T W_k,h / sqrt(d) for each head] ([heads*H padded][H]) with bias cross_kfc_b = W_q,h^T b_k,h / sqrt(d),
     * cross_vfc_w = [W_o[:,h] W_v,h for each head] with bias W_o[:,h] b_v,h, and for the per-position constant
     * cross_cu = W_k,h^T b_q,h / sqrt(d) ([heads][H] fp32), cross_c0 = b_k,h . b_q,h / sqrt(d) ([heads]).  With them
     * mic_tr_cross_kv is two GEMMs and one small launch per layer instead of five GEMMs, and never forms k and v. */
    const void *cross_kfc_w, *cross_vfc_w;
    const float *cross_kfc_b, *cross_vfc_b, *cross_cu, *cross_c0;
    /* optional, for at most MIC_FOLD_MAX_HEADS heads: the self-attention in-projection with the output projection
     * folded into the value rows, in fragment order: rows [W_q ; W_k ; W_o[:, head 0] W_v[head 0] ; ...] ((2+heads)H rows,
     * padded to 128), its bias [b_q ; b_k ; 0 ...] and self_of_b = b_o + sum_h W_o[:, head h] b_v[head h].  With them, the
     * folded cross weights and the workspace buffers qkvf / vfcache the step driver runs mic_attn_fused. */
    const void* self_qkvf_wf;
    const float *self_qkvf_b, *self_of_b;
} mic_tr_layer;

typedef struct mic_tr_decoder {
    int32_t dtype, H, heads, layers, vocab, pos_rows;
    const float* tok_emb;  /* fp32 [vocab][H] */
    const float* pos_emb;  /* fp32 [pos_rows][H] */
    mic_tr_layer layer[MIC_MAX_LAYERS];
    const void* out_w;     /* dtype [vocab padded][H] */
    const float* out_b;
    const void* out_wf;    /* optional: out_w in fragment order (see mic_tr_layer) */
} mic_tr_decoder;

/* Per-call decode workspace; every pointer is caller-allocated device memory of at least the
 * stated size (rows N = B * K, K = 1 for greedy / sampling). */
typedef struct mic_tr_workspace {
    int32_t N, t_cap, S, rows_per_image;
    float* x;        /* [N][H] residual stream */
    void* xt;        /* [N][H] dtype copy */
    float* y;        /* [N][H] */
    float* qkv;      /* [N][3H] */
    void* o;         /* [N][H] dtype */
    void* ffh;       /* [N][2H] dtype */
    float* logits;   /* [N][vocab] */
    void* kcache[MIC_MAX_LAYERS];   /* [N][t_cap][H] dtype */
    void* vcache[MIC_MAX_LAYERS];
    void* cross_kv[MIC_MAX_LAYERS]; /* [N/rows_per_image][S][2H] dtype */
    float* attn;     /* NULL, or beam: [N][t_cap][S]; sampling: [layers][t_cap][N][S] */
    float* top_val;  /* [N][K] */
    int32_t* top_idx;
    float* stats[2]; /* optional, each [N][ceil(H/128)][2]: row statistics for the LayerNorm-free step
                        (used when set, N <= 2048 and every fragment-ordered weight is present) */
    void* cross_kf[MIC_MAX_LAYERS];   /* optional, each [N/rows_per_image][S][heads][H] dtype: folded keys */
    void* cross_vf[MIC_MAX_LAYERS];   /*           same shape: folded values */
    float* cross_c[MIC_MAX_LAYERS];   /*           [N/rows_per_image][S][heads] fp32 */
    float* qkvf;                      /* optional, [N][(2+heads)H] fp32: packed q | k | folded values of the new position */
    void* vfcache[MIC_MAX_LAYERS];    /* optional, each [N][t_cap][heads][H] dtype: folded self-attention values */
    uint32_t* sync;                   /* optional, mic_tr_persist_sync_words(N, rows_per_image) words, zero when first used:
                                         seam counters and the error word of the one-launch step (mic_tr_step_persistent) */
} mic_tr_workspace;

/* Cross-attention keys/values of every layer from image tokens feats [B][S][H] (dtype):
 * the k/v thirds of cross_attention.mha.in_proj applied once per image instead of once per
 * beam per step (models/transformer.py:136 via :605). */
int mic_tr_cross_kv(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const void* feats, void* stream);

/* One decoder step for position t: embedding of tokens[n*ld_tok + t], every layer, vocabulary
 * projection into ws->logits (models/transformer.py:602-609 / :492-507 for the newest position).
 * anc: beam ancestry [N][t_cap] or NULL.  attn_mode: 0 none, 1 beam layout, 2 sampling layout. */
int mic_tr_step(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok,
                const int32_t* anc, int t, int attn_mode, void* stream);

/* mic_tr_step (the embedding aside, every layer and the vocabulary projection) as ONE launch: the rows of a step are
 * independent per image, so clusters of 16 workgroups (16 images each) carry their rows through every sub-layer and
 * exchange intermediate results inside the launch (write-through stores, a per-seam arrival counter, L1-bypassing
 * loads) instead of ending a kernel after every sub-layer.  Same arithmetic as mic_tr_step, bit for bit.  Replaces the
 * loop body of models/transformer.py:602-609 / :492-507 like mic_tr_step.  Needs what the fused-attention step needs
 * (folded weights, qkvf, vfcache, stats) plus ws->sync; supported shapes: mic_tr_persist_supported (one 16-byte
 * element per lane per row, at most 2 heads, 3 rows per image, 48 cached positions, 256 images).  The launch assumes
 * that its workgroups (one per CU, 16 per cluster) become resident together; every wait is bounded, and a wait
 * that runs out sets the error word ws->sync[mic_tr_persist_sync_words - 64] (non-zero = seam index + 0x100) and ends
 * the launch.  Do not run two of these launches concurrently on one GPU. */
int mic_tr_persist_supported(int dtype, int H, int heads, int layers, int S, int rows_per_image, int t_cap, int N);
int64_t mic_tr_persist_sync_words(int N, int rows_per_image);
int mic_tr_step_persistent(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok,
                           const int32_t* anc, int t, int attn_mode, void* stream);
/* Which schedule a step on (dec, ws) takes inside mic_tr_beam_steps / mic_tr_sample_steps: 3 = the one-launch step,
 * 2 = one launch per sub-layer with both attentions fused, 1 = chained skinny GEMMs, 0 = tiled GEMMs (no reference counterpart:
 * diagnostic, so that a benchmark line can say what it timed). */
int mic_tr_step_schedule(const mic_tr_decoder* dec, const mic_tr_workspace* ws);
/* Diagnostic, no reference counterpart: while `buffer` (device memory, 128 uint64 per workgroup of the launch: 512 * 128 * 8
 * bytes cover every shape) is set, every one-launch step writes 100 MHz time stamps of its phases into it: [0] start, then
 * per phase p: [3p+1] inputs may be read, [3p+2] outputs stored, [3p+3] arrival counted; GEMM phases also [40+6p .. 45+6p]: row statistics merged, rows loaded,
 * rows staged, matrix products done, epilogue values ready, output statistics stored.  NULL switches it off. */
int mic_tr_persist_trace(void* buffer);

/* mic_tr_step for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same step, same kernels,
 * same results as mic_tr_step; the difference is where the step's temporaries live: the fields x, xt, y, qkv,
 * o, ffh and stats[0..1] of `io` are IGNORED (may be NULL) and replaced by slices of a device array of
 * MIC_STATIC_SCRATCH_BYTES that is part of the library's code object (each GPU the library runs on holds its
 * own copy until the process ends).  Every other field of `io` (sizes, logits, caches, cross keys/values,
 * attn) is the caller's memory as in mic_tr_step.  The call allocates nothing; the first call on a device
 * looks the array's address up.  The temporaries take, each rounded up to 256 bytes,
 *     N*H*(4 + e + 4 + 12 + e + 2e) + 2 * N*ceil(H/128)*8 bytes   (e = element size of dec->dtype),
 * or, when the decoder carries the folded self-attention matrices and the fused attention launch applies (the
 * fields qkvf and vfcache of `io` must then be usable as for mic_tr_step; qkvf is replaced by a slice as well),
 *     N*H*(4 + e + 4 + 4*(2 + heads) + 2e) + 2 * N*ceil(H/128)*8 bytes,
 * and a call that needs more than MIC_STATIC_SCRATCH_BYTES is rejected with MIC_ERR_ARG (512 rows of hidden
 * size 512 in bf16 fit).  All calls on one GPU share that GPU's array, whatever their stream: calls that
 * overlap in time on one GPU race.  Nothing in the array has to start at zero (each temporary is written by the
 * step before it is read), so consecutive calls of any sizes need no clearing in between. */
#define MIC_STATIC_SCRATCH_BYTES (8u << 20)
int mic_tr_step_static(const mic_tr_decoder* dec, const mic_tr_workspace* io, const int32_t* tokens, int ld_tok,
                       const int32_t* anc, int t, int attn_mode, void* stream);

/* Steps [t0, t1) of transformer beam search on device: mic_tr_step, log-softmax top-K, beam
 * selection, ping-ponging between st[0] and st[1] (st[t & 1] is current at step t).
 * live_images: int32[t_cap] device counters, entry t receives the number of images with a live beam
 * after step t.  models/transformer.py:589-632.
 * use_graph != 0: the run is captured into a HIP graph the first time a given argument set is seen
 * (every pointer and size is part of the key) and replayed afterwards; the library keeps the 32 most
 * recently launched executables and frees a dropped one once its last launch has finished.  Capture instantiates a graph, i.e. allocates inside the runtime, on that first call. */
int mic_tr_beam_steps(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const mic_beam_state st[2],
                      int32_t* parent, int32_t* live_images, const double* sqrt_tab, int B, int K,
                      int t0, int t1, int eos_id, int want_attn, int use_graph, void* stream);

/* Steps [t0, t1) of batched greedy (u == NULL) or temperature sampling (u = uniforms [t_cap][N]).
 * models/transformer.py:490-547. */
int mic_tr_sample_steps(const mic_tr_decoder* dec, const mic_tr_workspace* ws, int32_t* tokens,
                        uint8_t* finished, float* total, int32_t* live_rows, float inv_temp, const float* u,
                        int t0, int t1, int pad_id, int eos_id, int want_attn, int use_graph, void* stream);

/* Executables the graph cache of the *_steps drivers holds (at most 32, least recently launched dropped first) and
 * dropped ones still parked because their last launch has not finished.  Diagnostic; no reference counterpart. */
int mic_graph_cache_stats(int32_t* cached, int32_t* parked);

typedef struct mic_lstm_decoder {
    int32_t dtype, E, Hd, layers, vocab, kind; /* kind 0 = basic, 1 = intermediate */
    int32_t kpad[MIC_MAX_LAYERS];              /* padded K = in_dim + Hd of each layer's GEMM */
    const float* embed;                        /* fp32 [vocab][E] */
    const void* w[MIC_MAX_LAYERS];             /* dtype [4Hd padded][kpad]: [W_ih | W_hh] */
    const float* b[MIC_MAX_LAYERS];            /* fp32 [4Hd] = b_ih + b_hh */
    const float *ln_g, *ln_b;                  /* intermediate only */
    const void* out_w;                         /* dtype [vocab padded][Hd padded] */
    const float* out_b;
    int32_t out_k;                             /* padded K of the vocabulary GEMM */
    /* optional: the same matrices in MFMA fragment order (mic_pack_weight_frag), rows zero-padded to 128; decode steps
     * of at most 2048 rows then take the weight-streaming GEMM (mic_linear_skinny) instead of the tiled one */
    const void* wf[MIC_MAX_LAYERS];
    const void* out_wf;
    /* optional: wf and b of a layer with rows in unit-major order (row 4u + g): mic_lstm_beam_steps then runs the layer's
     * cell in the epilogue of its gate GEMM (mic_lstm_gates_cell) where the workspace has the staging rows for h' (the
     * intermediate model's LayerNorm then follows the top layer as mic_layernorm) */
    const void* wf_cell[MIC_MAX_LAYERS];
    const float* b_cell[MIC_MAX_LAYERS];
} mic_lstm_decoder;

typedef struct mic_lstm_workspace {
    int32_t N;
    void* xh[MIC_MAX_LAYERS];   /* [N][kpad] dtype: [layer input | h] */
    float* h[MIC_MAX_LAYERS];   /* [N][Hd] */
    float* c[MIC_MAX_LAYERS];
    float* gates;               /* [N][4Hd] */
    float* hn;                  /* [N][Hd] LayerNorm output */
    void* hnt;                  /* [N][out_k] dtype */
    float* logits;              /* [N][vocab] */
    float* scratch;             /* [N][max(E,Hd)] fp32 */
    void* hstage[MIC_MAX_LAYERS]; /* optional [N][kpad] dtype, laid out like xh: the recurrent half receives h' of a layer whose
                                   * cell ran in the gate GEMM's epilogue (that launch is still reading xh); the beam tail
                                   * gathers the survivors' rows from here */
} mic_lstm_workspace;

/* One LSTM decoder step (all layers + LayerNorm (intermediate) + vocabulary projection) from the
 * carried (h, c).  If feat_t != NULL the layer-0 input is that row of image features (dtype,
 * [N][ld_feat]) instead of an embedded token: time-step 0 of models/basic.py:45-48.  Otherwise the
 * input is embed[tokens[n*ld_tok + pos]].  models/intermediate.py:59-66. */
int mic_lstm_step(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, const int32_t* tokens, int ld_tok,
                  int pos, const void* feat_t, int ld_feat, void* stream);

/* Steps [t0, t1) of LSTM beam search (models/image_captioner.py:118-159).  Step t runs on ws[t & 1]; the
 * survivors' c and [x | h] rows are gathered by parent into ws[(t + 1) & 1]; st as in mic_tr_beam_steps
 * (anc unused), seq rows hold t_cap tokens.  top_val/top_idx: [B*K][K] scratch.  With use_graph the run is
 * captured once per argument set and replayed. */
int mic_lstm_beam_steps(const mic_lstm_decoder* dec, const mic_lstm_workspace ws[2], const mic_beam_state st[2],
                        int32_t* parent, int32_t* live_images, const double* sqrt_tab, float* top_val, int32_t* top_idx,
                        int B, int K, int t0, int t1, int t_cap, int eos_id, int use_graph, void* stream);

/* Steps [t0, t1) of batched greedy (u == NULL) or temperature sampling (u = uniforms [steps][N]) with an
 * LSTM decoder (models/image_captioner.py:82-105).  val/idx: [N] scratch; tokens [N][t_cap]. */
int mic_lstm_sample_steps(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, int32_t* tokens, uint8_t* finished,
                          int32_t* live_rows, float inv_temp, const float* u, float* val, int32_t* idx, int t0, int t1,
                          int t_cap, int pad_id, int eos_id, int use_graph, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIC_HIP_H */
