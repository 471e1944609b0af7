// Pieces shared by the implicit-GEMM convolution kernels (igemm.hip: the generic tile variants; igemm8.hip: the
// 256 x 256 eight-phase variant): the launch arguments, the LDS tile image, the MFMA wrappers, the direct-to-LDS load
// and the per-chunk epilogue arithmetic, so that every variant rounds a result the same way.
#pragma once
#include "mic_common.h"

struct IgemmArgs {
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    int M, N, K;
    int OHW, OW, IH, IW;
    long long img_stride, row_stride;
    int pix_stride, kw_taps, run, stride, pad_h, pad_w;
    int ldc, ld_res, relu, out_dtype, res_dtype;
    int tiles_m, tiles_n, nk, vec_ok;
    // optional second K segment: k in [K, Kt) reads channel k-K of in2 at pixel (oh*stride2, ow*stride2)
    const void* in2;
    int K2, Kt;
    long long img_stride2, row_stride2;
    int pix_stride2, stride2;
    // LayerNorm epilogue (LN variants): out = res + LayerNorm(acc + bias) * ln_g + ln_b in fp32, ln_yt = the same in T
    const float* ln_g;
    const float* ln_b;
    void* ln_yt;
    int ld_yt;
    // split-K scratch of the 256 x 256 eight-phase variant: tickets (zero between launches) then fp32 partial tiles
    void* scratch;
    long long scratch_bytes;
    int splitk;
};

namespace {

// LDS tile image: rows of 64 bytes (4 chunks of 16 B).  Chunk q of row r is stored at chunk
// q ^ (2 * bit3(r)): with that, the four 16-lane groups a ds_read_b128 is served in each touch 16
// distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int lds_off(int row, int q) { return row * 64 + ((q ^ (((row >> 3) & 1) << 1)) << 4); }

// x/2 * (1 + erf(x / sqrt 2)) with erf from the rational-exponential form of Abramowitz & Stegun 7.1.26 (absolute error
// below 1.5e-7, i.e. fp32 rounding of the result), branch-free: erfc(|z|) = poly(t) * exp(-z^2), t = 1 / (1 + p |z|), and
// 1 + erf(z) is erfc(|z|) for z < 0 (no cancellation in the left tail) and 2 - erfc(|z|) otherwise.  The library erff is a
// piecewise polynomial with divergent branches and costs several times as much in an epilogue that is otherwise memory bound.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_abs = p * t * __expf(-z * z);
    return 0.5f * x * (x < 0.f ? erfc_abs : 2.0f - erfc_abs);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // 16x16x4 f32 MFMA: lane l supplies k = l>>4.  Each lane holds 4 consecutive k of its chunk, so
    // MFMA e covers k = {4q+e}: the four together sum the whole 16-element k-tile.
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        const f32x4_t a = __builtin_bit_cast(f32x4_t, wf), b = __builtin_bit_cast(f32x4_t, xf);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
};

// RES_PRE: the bf16 residual tile is requested before the main loop and held in registers, so its HBM
// latency hides behind the loads and MFMAs instead of sitting in the epilogue (the 1x1 expansion
// convolutions have K as small as 64: their time is the residual read and the output write).
//
// GLDS: operand tiles go global -> LDS directly (global_load_lds_dwordx4, no register staging and no
// ds_write pass).  One wave-instruction fills 1 KiB = 16 rows of one 64-byte sub-tile in lane order;
// the XOR swizzle of the LDS image is applied to the SOURCE chunk each lane fetches.  Rows outside the
// problem, taps outside the image and the K tail fetch a 16-byte block of zeros instead.
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* lptr_t;

// One global_load_lds_dwordx4: every lane fetches 16 bytes from its own global address, the wave's
// 1 KiB lands at LDS address `lds_base` + 16 * lane.  Issued from inline assembly on purpose: the
// compiler tracks a direct-to-LDS load it can see as a pending LDS write and drains the vector-memory
// counter (vmcnt(0)) before the next LDS read, which would serialise the multi-stage pipeline.  Ordering
// is explicit instead: counted s_waitcnt vmcnt + s_barrier before a stage is read.  M0 (the LDS base
// register) belongs to the compiler, so it is saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

// Epilogue arithmetic of one chunk of 8 consecutive output channels of one pixel: v holds the fp32 accumulators; adds
// the bias and the residual (`pre`: the residual chunk when it was requested ahead of the main loop), applies the
// activation and stores the chunk.  `full`: the chunk lies inside N and every access is 16-byte aligned.
template <bool RES_PRE> __device__ __forceinline__ void igemm_finish_chunk(const IgemmArgs& a, float (&v)[8], int m, int n, const uint4& pre) {
    const bool full = a.vec_ok && n + 7 < a.N;
    if (a.bias) {
        if (full) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n), b1 = *reinterpret_cast<const float4*>(a.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < a.N) v[e] += a.bias[n + e];
        }
    }
    if (RES_PRE && full) {
        const Vec16<bf16_t> t = __builtin_bit_cast(Vec16<bf16_t>, pre);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)t.v[e];
    } else if (a.res) {
        const long long ro = (long long)m * a.ld_res + n;
        if (a.res_dtype == MIC_F32) {
            const float* rp = (const float*)a.res + ro;
            if (full) {
                const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < a.N) v[e] += rp[e];
            }
        } else {
            const bf16_t* rp = (const bf16_t*)a.res + ro;
            if (full) {
                const Vec16<bf16_t> t = ld16(rp);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)t.v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < a.N) v[e] += (float)rp[e];
            }
        }
    }
    if (a.relu == MIC_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (a.relu == MIC_ACT_GELU) {  // exact (erf) form, nn.GELU() default
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    }
    const long long oo = (long long)m * a.ldc + n;
    if (a.out_dtype == MIC_F32) {
        float* o = (float*)a.out + oo;
        if (full) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < a.N) o[e] = v[e];
        }
    } else {
        bf16_t* o = (bf16_t*)a.out + oo;
        if (full) {
            Vec16<bf16_t> p;
#pragma unroll
            for (int e = 0; e < 8; ++e) p.v[e] = (bf16_t)v[e];
            st16(o, p);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < a.N) o[e] = (bf16_t)v[e];
        }
    }
}

}  // namespace

// Whether every operand of the layer can be addressed by a 32-bit byte offset into a 2 GiB buffer descriptor with one
// validity bit per tap: tensors below 2 GiB as far as the descriptor says (the last image's last tap's last channel), at
// most 31 taps.
static inline long long igemm_extent_bytes(long long images, long long img_stride, long long rows, long long row_stride, long long cols,
                                           long long pix_stride, int run, int elem) {
    return ((images - 1) * img_stride + (rows - 1) * row_stride + (cols - 1) * pix_stride + run) * elem;
}
static inline bool igemm_fits_buffers(const IgemmArgs& a, int elem) {
    if (a.K / a.run > 31) return false;
    const long long lim = (1ll << 31) - (1 << 20);
    const long long images = (a.M + a.OHW - 1) / a.OHW;
    if (igemm_extent_bytes(images, a.img_stride, a.IH, a.row_stride, a.IW, a.pix_stride, a.run, elem) > lim) return false;
    if ((long long)((a.N + 127) / 128 * 128) * a.Kt * elem > lim) return false;
    if (a.in2) {
        const int oh = a.OHW / a.OW;
        if (igemm_extent_bytes(images, a.img_stride2, (long long)(oh - 1) * a.stride2 + 1, a.row_stride2, (long long)(a.OW - 1) * a.stride2 + 1,
                               a.pix_stride2, a.K2, elem) > lim)
            return false;
    }
    return true;
}

// the eight-phase 256 x 256 bf16 variant (igemm8.hip): whether it takes the problem, and its launch
bool mic_igemm8_supported(const IgemmArgs& a);
int mic_igemm8_splitk(const IgemmArgs& a);  // slices per tile the launch would use (1: no split)
int mic_igemm8_launch(IgemmArgs& a, hipStream_t s, double flops);
