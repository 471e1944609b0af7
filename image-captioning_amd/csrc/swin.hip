// Swin-V2 encoder: the kernels that are not GEMMs (4x4 patch gather, shifted-window cosine
// attention on the matrix cores) and the native driver that chains them with the implicit-GEMM
// kernel and the LayerNorm kernel into swin_v2_*.features (models/encoders/swin.py:68).
//
// Token layout: the channels-last map (B, H, W, C) of each stage is the row-major token matrix
// [B*H*W][C].  The residual stream is fp32; every GEMM input is a copy in the compute dtype.
// Windows, the cyclic shift and the zero padding to a multiple of the window never materialise:
// the attention kernel maps window slots to source tokens and substitutes the projection bias for
// padded slots (a zero input row projects to the bias).
#include "mic_common.h"

#include <memory>

namespace {

constexpr int WIN = 8;          // window side
constexpr int WTOK = WIN * WIN; // tokens per window
constexpr int HD = 32;          // channels per head in every Swin configuration

// ---------------------------------------------------------------------------------------------
// (B,3,H,W) fp32 -> [B][H/4][W/4][64] patches, k = (kh*4 + kw)*4 + c with a zero 4th channel.
// One thread per (patch, kernel row): 3 x 4 contiguous pixels in, 16 elements out.
template <typename T>
__global__ void patchify4_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int PH, int PW) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * PH * PW * 4;
    if (idx >= total) return;
    const int kh = (int)(idx & 3);
    const long long p = idx >> 2;
    const int pw = (int)(p % PW), ph = (int)((p / PW) % PH), b = (int)(p / ((long long)PW * PH));
    float v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* src = img + (((long long)b * 3 + c) * H + ph * 4 + kh) * W + pw * 4;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) v[c][kw] = src[kw];
    }
    T* o = out + p * 64 + kh * 16;
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
        o[kw * 4 + 0] = from_f32<T>(v[0][kw]);
        o[kw * 4 + 1] = from_f32<T>(v[1][kw]);
        o[kw * 4 + 2] = from_f32<T>(v[2][kw]);
        o[kw * 4 + 3] = from_f32<T>(0.f);
    }
}

// ---------------------------------------------------------------------------------------------
struct WinArgs {
    const void* qkv;            // [B*H*W][3C]
    const float* qkv_bias;      // [3C], key third zero
    const float* bias_tab;      // [heads][64][64]
    const float* logit_scale;   // [heads], already exp(min(., log 100))
    void* o;                    // [B*H*W][C]
    int B, H, W, C, heads, pH, pW, sh, sw, nWw, nW;
};

template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) {
    const Vec16<bf16_t> t = ld16(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t.v[e];
}

__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& lo, const f32x4_t& hi) {
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (bf16_t)lo[e]; r[4 + e] = (bf16_t)hi[e]; }
    return r;
}

// One workgroup (4 waves) per (image, window, head); wave w owns queries 16w..16w+15.
//   S^T = K Q^T   (MFMA A = keys, B = queries): lane holds its query (lane & 15) against keys
//                  16*kt + 4*(lane >> 4) + r, so a softmax row lives in 4 lanes x 16 registers;
//   O^T = V^T P^T (MFMA A = values transposed, B = probabilities): the probability registers are
//                  already in B-operand order once the k index of the MFMA is mapped to
//                  key = 16*kt + 4*(lane >> 4) + r, which only fixes how V^T is read from LDS.
template <typename T>
__global__ __launch_bounds__(256) void window_attn_kernel(const WinArgs a) {
    constexpr int LQ = HD + 4, LV = WTOK + 4;
    __shared__ __attribute__((aligned(16))) float sQ[WTOK * LQ];
    __shared__ __attribute__((aligned(16))) float sK[WTOK * LQ];
    __shared__ __attribute__((aligned(16))) float sVt[HD * LV];
    __shared__ int sReg[WTOK];
    __shared__ int sRow[WTOK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int blk = blockIdx.x;
    const int h = blk % a.heads;
    blk /= a.heads;
    const int win = blk % a.nW, b = blk / a.nW;
    const int wrow = win / a.nWw, wcol = win - wrow * a.nWw;

    {   // stage q (normalised), k (normalised), v^T of this head
        const int tok = tid >> 2, qd = tid & 3;
        const int i = wrow * WIN + (tok >> 3), j = wcol * WIN + (tok & 7);  // position in the rolled, padded map
        int si = i + a.sh, sj = j + a.sw;                                     // where the roll took it from
        if (si >= a.pH) si -= a.pH;
        if (sj >= a.pW) sj -= a.pW;
        const bool real = si < a.H && sj < a.W;
        const int row = (b * a.H + si) * a.W + sj;
        const int c0 = h * HD + qd * 8;
        float q[8], k[8], v[8];
        if (real) {
            const T* p = (const T*)a.qkv + (long long)row * 3 * a.C + c0;
            load8<T>(p, q);
            load8<T>(p + a.C, k);
            load8<T>(p + 2 * a.C, v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q[e] = a.qkv_bias[c0 + e];
                k[e] = a.qkv_bias[a.C + c0 + e];
                v[e] = a.qkv_bias[2 * a.C + c0 + e];
            }
        }
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sq += q[e] * q[e]; sk += k[e] * k[e]; }
        sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64);
        sk += __shfl_xor(sk, 1, 64); sk += __shfl_xor(sk, 2, 64);
        const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);  // F.normalize eps
        float* dq = sQ + tok * LQ + qd * 8;
        float* dk = sK + tok * LQ + qd * 8;
        *reinterpret_cast<float4*>(dq) = make_float4(q[0] * iq, q[1] * iq, q[2] * iq, q[3] * iq);
        *reinterpret_cast<float4*>(dq + 4) = make_float4(q[4] * iq, q[5] * iq, q[6] * iq, q[7] * iq);
        *reinterpret_cast<float4*>(dk) = make_float4(k[0] * ik, k[1] * ik, k[2] * ik, k[3] * ik);
        *reinterpret_cast<float4*>(dk + 4) = make_float4(k[4] * ik, k[5] * ik, k[6] * ik, k[7] * ik);
#pragma unroll
        for (int e = 0; e < 8; ++e) sVt[(qd * 8 + e) * LV + tok] = v[e];
        if (qd == 0) {
            sRow[tok] = real ? row : -1;
            // regions of the shift mask: [0, n-8), [n-8, n-shift), [n-shift, n) along each shifted axis
            const int hr = a.sh == 0 ? 0 : (i < a.pH - WIN ? 0 : (i < a.pH - a.sh ? 1 : 2));
            const int wr = a.sw == 0 ? 0 : (j < a.pW - WIN ? 0 : (j < a.pW - a.sw ? 1 : 2));
            sReg[tok] = hr * 3 + wr;
        }
    }
    __syncthreads();

    const int ql = wave * 16 + (lane & 15), g = lane >> 4, lr = lane & 15;
    f32x4_t s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
        const float* qp = sQ + ql * LQ + 8 * g;
        const bf16x8_t bq = pack8(*reinterpret_cast<const f32x4_t*>(qp), *reinterpret_cast<const f32x4_t*>(qp + 4));
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float* kp = sK + (16 * kt + lr) * LQ + 8 * g;
            const bf16x8_t ak = pack8(*reinterpret_cast<const f32x4_t*>(kp), *reinterpret_cast<const f32x4_t*>(kp + 4));
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bq, s[kt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4_t bq = *reinterpret_cast<const f32x4_t*>(sQ + ql * LQ + 16 * half + 4 * g);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4_t ak = *reinterpret_cast<const f32x4_t*>(sK + (16 * kt + lr) * LQ + 16 * half + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[e], bq[e], s[kt], 0, 0, 0);
            }
        }
    }

    // logits = cos * scale + position bias + shift mask; softmax over the 64 keys of the query
    const float scale = a.logit_scale[h];
    const float* bt = a.bias_tab + ((long long)h * WTOK + ql) * WTOK;
    const int regq = sReg[ql];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const float4 bb = *reinterpret_cast<const float4*>(bt + 16 * kt + 4 * g);
        const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = sReg[16 * kt + 4 * g + r] != regq ? -100.f : 0.f;
            s[kt][r] = s[kt][r] * scale + bv[r] + m;
            mx = fmaxf(mx, s[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = expf(s[kt][r] - mx);
            sum += s[kt][r];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] *= inv;

    f32x4_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const float* vp = sVt + (16 * dt + lr) * LV + 4 * g;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const bf16x8_t av = pack8(*reinterpret_cast<const f32x4_t*>(vp + 32 * t2), *reinterpret_cast<const f32x4_t*>(vp + 32 * t2 + 16));
                const bf16x8_t bp = pack8(s[2 * t2], s[2 * t2 + 1]);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bp, o[dt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4_t av = *reinterpret_cast<const f32x4_t*>(vp + 16 * kt);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], s[kt][e], o[dt], 0, 0, 0);
            }
        }
    }
    const int row = sRow[ql];
    if (row >= 0) {
        T* op = (T*)a.o + (long long)row * a.C + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) op[16 * dt + r] = from_f32<T>(o[dt][r]);
    }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int linear(int dtype, const void* x, int M, int K, const void* w, const float* bias, void* out, int out_dtype, int N, int act,
           hipStream_t s) {
    mic_conv_desc d{};
    d.in = x; d.w = w; d.bias = bias; d.out = out;
    d.dtype = dtype; d.out_dtype = out_dtype; d.res_dtype = MIC_F32;
    d.M = M; d.N = N; d.K = K;
    d.OH = d.OW = d.IH = d.IW = 1;
    d.in_img_stride = K; d.in_pix_stride = 4;
    d.kw_taps = 1; d.run = K; d.stride = 1;
    d.ldc = N; d.relu = act;
    return mic_conv2d(&d, s);
}

// y = LayerNorm(x W^T + bias) * g + be (+ res): the normalisation rides in the GEMM epilogue when one tile holds a whole
// row, else the fp32 product goes through `u` and the stand-alone kernel.  res may be the same buffer as y.
int linear_ln(int dtype, const void* x, int M, int K, const void* w, const float* bias, const float* g, const float* be,
              const float* res, float* y, void* yt, int N, float* u, bool fuse, hipStream_t s) {
    if (fuse && mic_conv2d_ln_supported(N)) {
        mic_conv_desc d{};
        d.in = x; d.w = w; d.bias = bias; d.out = y; d.res = res;
        d.dtype = dtype; d.out_dtype = MIC_F32; d.res_dtype = MIC_F32;
        d.M = M; d.N = N; d.K = K;
        d.OH = d.OW = d.IH = d.IW = 1;
        d.in_img_stride = K; d.in_pix_stride = 4;
        d.kw_taps = 1; d.run = K; d.stride = 1;
        d.ldc = N; d.ld_res = N; d.relu = MIC_ACT_NONE;
        d.ln_gamma = g; d.ln_beta = be; d.ln_out = yt; d.ld_ln_out = N;
        return mic_conv2d(&d, s);
    }
    MIC_TRY(linear(dtype, x, M, K, w, bias, u, MIC_F32, N, MIC_ACT_NONE, s));
    return res ? mic_layernorm_add(u, g, be, res, y, yt, N, dtype, M, N, s) : mic_layernorm(u, g, be, y, yt, N, dtype, M, N, s);
}

struct SwinPlan {
    int PH, PW;
    int64_t x, xt, qkv, o, u, h, total;
};

SwinPlan plan_swin(const mic_swin* net, int B, int H, int W) {
    SwinPlan p;
    const int64_t sz = (int64_t)mic_dtype_size(net->dtype);
    p.PH = H / 4; p.PW = W / 4;
    const int64_t M = (int64_t)B * p.PH * p.PW, C = net->embed_dim;  // stage 0 has the largest M*C
    p.x = align_up(M * C * 4, 256);
    p.xt = align_up(M * C * sz, 256);
    p.qkv = align_up(M * 3 * C * sz, 256);
    p.o = align_up(M * C * sz, 256);
    p.u = align_up(M * C * 4, 256);
    const int64_t hid = M * 4 * C, patch = M * 64;
    p.h = align_up((hid > patch ? hid : patch) * sz, 256);
    p.total = p.x + p.xt + p.qkv + p.o + p.u + p.h;
    return p;
}

int check_swin(const mic_swin* net, const char* who) {
    MIC_ARG_CHECK(net->dtype == MIC_F32 || net->dtype == MIC_BF16, "%s: bad dtype", who);
    MIC_ARG_CHECK(net->embed_dim > 0 && net->embed_dim % 32 == 0, "%s: embed_dim %d must be a multiple of 32", who, net->embed_dim);
    int blocks = 0;
    for (int s = 0; s < 4; ++s) {
        MIC_ARG_CHECK(net->depths[s] > 0, "%s: stage %d has no blocks", who, s);
        MIC_ARG_CHECK(net->heads[s] * HD == (net->embed_dim << s), "%s: stage %d needs %d heads of %d channels, got %d", who, s,
                      (net->embed_dim << s) / HD, HD, net->heads[s]);
        blocks += net->depths[s];
    }
    MIC_ARG_CHECK(blocks <= MIC_SWIN_MAX_BLOCKS, "%s: %d blocks (max %d)", who, blocks, MIC_SWIN_MAX_BLOCKS);
    return 0;
}

}  // namespace

extern "C" int mic_patchify4(const float* images, void* out, int dtype, int B, int H, int W, void* stream) {
    MIC_ARG_CHECK(images && out, "mic_patchify4: null pointer");
    MIC_ARG_CHECK(B > 0 && H >= 4 && W >= 4, "mic_patchify4: bad sizes B=%d H=%d W=%d", B, H, W);
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_patchify4: bad dtype %d", dtype);
    const int PH = H / 4, PW = W / 4;
    const long long total = (long long)B * PH * PW * 4;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(patchify4_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, images, (bf16_t*)out, B, H, W, PH, PW);
    else
        MIC_LAUNCH(patchify4_kernel<float>, dim3(grid), dim3(256), 0, s, images, (float*)out, B, H, W, PH, PW);
    return 0;
}

extern "C" int mic_window_attn(const void* qkv, const float* qkv_bias, const float* bias_tab, const float* logit_scale, void* o,
                               int dtype, int B, int H, int W, int C, int heads, int shift, void* stream) {
    MIC_ARG_CHECK(qkv && qkv_bias && bias_tab && logit_scale && o, "mic_window_attn: null pointer");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_window_attn: bad dtype %d", dtype);
    MIC_ARG_CHECK(B > 0 && H > 0 && W > 0, "mic_window_attn: bad sizes B=%d H=%d W=%d", B, H, W);
    MIC_ARG_CHECK(heads > 0 && C == heads * HD, "mic_window_attn: C=%d must be heads(%d) x %d", C, heads, HD);
    MIC_ARG_CHECK(shift >= 0 && shift < WIN, "mic_window_attn: shift %d outside [0,%d)", shift, WIN);
    MIC_ARG_CHECK((long long)B * H * W < (1ll << 31), "mic_window_attn: too many tokens");
    MIC_ARG_CHECK(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)bias_tab & 15) == 0, "mic_window_attn: pointers must be 16-byte aligned");
    WinArgs a;
    a.qkv = qkv; a.qkv_bias = qkv_bias; a.bias_tab = bias_tab; a.logit_scale = logit_scale; a.o = o;
    a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads;
    a.pH = (H + WIN - 1) / WIN * WIN; a.pW = (W + WIN - 1) / WIN * WIN;
    a.sh = WIN >= a.pH ? 0 : shift;  // one window already spans the axis: nothing to shift
    a.sw = WIN >= a.pW ? 0 : shift;
    a.nWw = a.pW / WIN; a.nW = (a.pH / WIN) * a.nWw;
    const long long grid = (long long)B * a.nW * heads;
    MIC_ARG_CHECK(grid < (1ll << 31), "mic_window_attn: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(window_attn_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else
        MIC_LAUNCH(window_attn_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return 0;
}

extern "C" int64_t mic_swin_workspace_bytes(const mic_swin* net, int B, int H, int W) {
    if (!net || B <= 0 || H < 32 || W < 32 || check_swin(net, "mic_swin_workspace_bytes") != 0) return -1;
    return plan_swin(net, B, H, W).total;
}

extern "C" int mic_swin_forward(const mic_swin* net, const float* images, int B, int H, int W, void* workspace,
                                int64_t workspace_bytes, void* out, void* stream) {
    MIC_ARG_CHECK(net && images && workspace && out, "mic_swin_forward: null pointer");
    MIC_ARG_CHECK(B > 0 && H >= 32 && W >= 32, "mic_swin_forward: bad image size %dx%d (B=%d)", H, W, B);
    MIC_TRY(check_swin(net, "mic_swin_forward"));
    MIC_ARG_CHECK(((uintptr_t)workspace & 255) == 0, "mic_swin_forward: workspace must be 256-byte aligned");
    const SwinPlan p = plan_swin(net, B, H, W);
    MIC_ARG_CHECK(workspace_bytes >= p.total, "mic_swin_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)p.total);
    MIC_ARG_CHECK((long long)B * p.PH * p.PW < (1ll << 31), "mic_swin_forward: batch too large for 32-bit row counts");
    hipStream_t s = (hipStream_t)stream;
    const int dt = net->dtype;
    char* base = (char*)workspace;
    float* x = (float*)base;
    void* xt = base + p.x;
    void* qkv = base + p.x + p.xt;
    void* o = base + p.x + p.xt + p.qkv;
    float* u = (float*)(base + p.x + p.xt + p.qkv + p.o);
    void* hbuf = base + p.x + p.xt + p.qkv + p.o + p.u;

    int h = p.PH, w = p.PW, C = net->embed_dim;
    MIC_ARG_CHECK(net->patch_w && net->patch_b && net->patch_g && net->patch_be, "mic_swin_forward: patch embedding not set");
    MIC_TRY(mic_patchify4(images, hbuf, dt, B, H, W, s));
    // the GEMMs between two window-attention launches run back to back: one pair of profiling events per such run
    const int prof_slot = dt == MIC_BF16 ? 1 : 0;
    std::unique_ptr<MicProfScope> scope(new MicProfScope(s, prof_slot));
    static const bool fuse_ln = [] {
        const char* e = getenv("MIC_SWIN_NO_LN_FUSE");  // tuning / comparison runs: LayerNorm as its own launch
        return !(e && atoi(e));
    }();
    MIC_TRY(linear_ln(dt, hbuf, B * h * w, 64, net->patch_w, net->patch_b, net->patch_g, net->patch_be, nullptr, x, xt, C, u, fuse_ln, s));
    int bi = 0;
    for (int st = 0; st < 4; ++st) {
        const int M = B * h * w, heads = net->heads[st];
        for (int k = 0; k < net->depths[st]; ++k, ++bi) {
            const mic_swin_block& Bk = net->block[bi];
            MIC_ARG_CHECK(Bk.qkv_w && Bk.proj_w && Bk.fc1_w && Bk.fc2_w && Bk.qkv_b && Bk.bias_tab && Bk.logit_scale,
                          "mic_swin_forward: block %d not set", bi);
            const bool last = st == 3 && k == net->depths[3] - 1;
            MIC_TRY(linear(dt, xt, M, C, Bk.qkv_w, Bk.qkv_b, qkv, dt, 3 * C, MIC_ACT_NONE, s));
            scope.reset();
            MIC_TRY(mic_window_attn(qkv, Bk.qkv_b, Bk.bias_tab, Bk.logit_scale, o, dt, B, h, w, C, heads, (k & 1) ? WIN / 2 : 0, s));
            scope.reset(new MicProfScope(s, prof_slot));
            MIC_TRY(linear_ln(dt, o, M, C, Bk.proj_w, Bk.proj_b, Bk.n1_g, Bk.n1_b, x, x, xt, C, u, fuse_ln, s));
            MIC_TRY(linear(dt, xt, M, C, Bk.fc1_w, Bk.fc1_b, hbuf, dt, 4 * C, MIC_ACT_GELU, s));
            MIC_TRY(linear_ln(dt, hbuf, M, 4 * C, Bk.fc2_w, Bk.fc2_b, Bk.n2_g, Bk.n2_b, x, x, last ? out : xt, C, u, fuse_ln, s));
        }
        if (st < 3) {  // 2x2 neighbourhood -> 4C -> 2C as a 2x2/2 convolution, then LayerNorm
            const mic_swin_merge& Mg = net->merge[st];
            MIC_ARG_CHECK(Mg.w && Mg.g && Mg.b, "mic_swin_forward: merge %d not set", st);
            const int oh = (h + 1) / 2, ow = (w + 1) / 2;
            mic_conv_desc d{};
            d.in = xt; d.w = Mg.w; d.out = u;
            d.dtype = dt; d.out_dtype = MIC_F32; d.res_dtype = MIC_F32;
            d.M = B * oh * ow; d.N = 2 * C; d.K = 4 * C;
            d.OH = oh; d.OW = ow; d.IH = h; d.IW = w;
            d.in_img_stride = (int64_t)h * w * C; d.in_row_stride = (int64_t)w * C; d.in_pix_stride = C;
            d.kw_taps = 2; d.run = C; d.stride = 2;
            d.ldc = 2 * C;
            const bool fused = fuse_ln && mic_conv2d_ln_supported(2 * C);
            if (fused) {  // the launch reads xt while it writes the normalised rows: they go to the idle attention buffer
                d.out = x;
                d.ln_gamma = Mg.g; d.ln_beta = Mg.b; d.ln_out = o; d.ld_ln_out = 2 * C;
            }
            MIC_TRY(mic_conv2d(&d, s));
            if (fused) { void* t = xt; xt = o; o = t; }
            h = oh; w = ow; C *= 2;
            if (!fused) MIC_TRY(mic_layernorm(u, Mg.g, Mg.b, x, xt, C, dt, B * h * w, C, s));
        }
    }
    return 0;
}
