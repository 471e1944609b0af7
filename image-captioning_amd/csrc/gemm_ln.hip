// Skinny fused layer tail:  y = LayerNorm(x W^T + bias + res) * gamma + beta   (eps 1e-5)
//
// One workgroup owns 16 complete rows, so the row statistics never leave the block and the three
// kernels (GEMM, residual add, LayerNorm) of a post-LN transformer sub-layer become one launch.
// With one new token per beam the row count is small (hundreds), so nothing is shared between
// workgroups but the weights, and those are streamed straight from L2 into MFMA fragments.  The
// weights are pre-packed in fragment order (mic_pack_weight_frag): the 1 KiB that the 64 lanes of a
// wave need for one 16-row x 64-byte tile is contiguous, lane-major, so each fragment is one fully
// coalesced 16-byte-per-lane load.  (Reading the row-major matrix instead makes every lane of a
// load touch a different cache line and the kernel runs 5x slower on address processing alone.)
// The 16 activation rows are staged once through LDS; there is no barrier in the K loop, only the
// staging barrier and the two row reductions across waves.
#include "mic_common.h"

namespace {

template <typename T> struct SkinnyMma;
template <> struct SkinnyMma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf), acc, 0, 0, 0);
    }
};
template <> struct SkinnyMma<float> {
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        const f32x4_t a = __builtin_bit_cast(f32x4_t, wf), b = __builtin_bit_cast(f32x4_t, xf);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
};

struct GemmLnArgs {
    const void* x;
    const void* w;
    const float *bias, *res, *gamma, *beta;
    float* y;
    void* yt;
    int ld_x, ld_res, ld_yt, M, K, H;
};

// NW waves, each TPW 16-column tiles: H == 16 * NW * TPW
template <typename T, int NW, int TPW>
__global__ __launch_bounds__(NW * 64) void gemm_res_ln_kernel(const GemmLnArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T), EPR = 4 * VEC;
    __shared__ float red[2][NW][16];
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];  // 16 rows of K elements, rows padded by 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int m = m0 + r;
    const int mc = m < a.M ? m : a.M - 1;
    const int nb = wave * TPW * 16;
    const int nk = a.K / EPR;
    const int xrow = a.K * (int)sizeof(T) + 16;
    {   // stage the activation rows: consecutive threads read consecutive 16-byte chunks of a row
        const int cpr = a.K / VEC;
        for (int c = tid; c < 16 * cpr; c += NW * 64) {
            const int row = c / cpr, ch = c - row * cpr;
            const int mr = m0 + row < a.M ? m0 + row : a.M - 1;
            *reinterpret_cast<uint4*>(xs + row * xrow + ch * 16) = *reinterpret_cast<const uint4*>((const T*)a.x + (long long)mr * a.ld_x + ch * VEC);
        }
    }
    const unsigned char* xp = xs + r * xrow + q * 16;
    // fragment-packed weights: [n_tile][k_tile][lane][VEC]
    const T* wp = (const T*)a.w + ((long long)(nb / 16) * nk * 64 + lane) * VEC;
    __syncthreads();

    f32x4_t acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Software pipeline over groups of U k-tiles with two register sets: the loads of group g+1 are
    // issued before the MFMAs of group g, so U * (TPW + 1) 16-byte loads per lane stay in flight
    // while the matrix core works (the operands come from L2 / Infinity Cache: latency bound).
    constexpr int U = TPW >= 8 ? 2 : 4;
    uint4 xa[U], wa[U][TPW], xb[U], wb[U][TPW];
    auto load_group = [&](int kt0, uint4 (&xf)[U], uint4 (&wf)[U][TPW]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt0 + u < nk ? kt0 + u : nk - 1;  // past the end: re-read the last tile, never used
            xf[u] = *reinterpret_cast<const uint4*>(xp + kt * 64);
#pragma unroll
            for (int j = 0; j < TPW; ++j) wf[u][j] = *reinterpret_cast<const uint4*>(wp + ((long long)j * nk + kt) * 64 * VEC);
        }
    };
    auto mma_group = [&](int kt0, const uint4 (&xf)[U], const uint4 (&wf)[U][TPW]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt0 + u < nk) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) SkinnyMma<T>::run(wf[u][j], xf[u], acc[j]);
            }
        }
    };
    load_group(0, xa, wa);
    for (int kt = 0; kt < nk; kt += 2 * U) {
        load_group(kt + U, xb, wb);
        mma_group(kt, xa, wa);
        load_group(kt + 2 * U, xa, wa);
        mma_group(kt + U, xb, wb);
    }
    // lane holds columns nb + j*16 + q*4 + e of row m
    float v[TPW][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = nb + j * 16 + q * 4;
        const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 rs = a.res ? *reinterpret_cast<const float4*>(a.res + (long long)mc * a.ld_res + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[j][0] = acc[j][0] + b.x + rs.x;
        v[j][1] = acc[j][1] + b.y + rs.y;
        v[j][2] = acc[j][2] + b.z + rs.z;
        v[j][3] = acc[j][3] + b.w + rs.w;
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) red[0][wave][r] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[0][w][r];
    const float mean = tot / (float)a.H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] -= mean;
            sq = fmaf(v[j][e], v[j][e], sq);
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (q == 0) red[1][wave][r] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) var += red[1][w][r];
    const float rstd = 1.0f / sqrtf(var / (float)a.H + 1e-5f);
    if (m >= a.M) return;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = nb + j * 16 + q * 4;
        const float4 g = *reinterpret_cast<const float4*>(a.gamma + n), b = *reinterpret_cast<const float4*>(a.beta + n);
        float4 o;
        o.x = v[j][0] * rstd * g.x + b.x;
        o.y = v[j][1] * rstd * g.y + b.y;
        o.z = v[j][2] * rstd * g.z + b.z;
        o.w = v[j][3] * rstd * g.w + b.w;
        if (a.y) *reinterpret_cast<float4*>(a.y + (long long)m * a.H + n) = o;
        if (a.yt) {
            T* t = (T*)a.yt + (long long)m * a.ld_yt + n;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(t) = o;
            } else {
                struct alignas(8) B4 { bf16_t v[4]; } p;
                p.v[0] = (bf16_t)o.x; p.v[1] = (bf16_t)o.y; p.v[2] = (bf16_t)o.z; p.v[3] = (bf16_t)o.w;
                *reinterpret_cast<B4*>(t) = p;
            }
        }
    }
}

template <typename T, int NW, int TPW> int launch_gl(const GemmLnArgs& a, hipStream_t s) {
    const size_t lds = 16 * ((size_t)a.K * sizeof(T) + 16);
    MIC_ARG_CHECK(lds <= 150 * 1024, "mic_linear_res_ln: K=%d needs %zu bytes of LDS", a.K, lds);
    if (lds > 48 * 1024) {
        static size_t granted = 0;  // raise the dynamic-LDS limit of this variant once per size class
        if (lds > granted) {
            MIC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_res_ln_kernel<T, NW, TPW>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            granted = 150 * 1024;
        }
    }
    MIC_LAUNCH((gemm_res_ln_kernel<T, NW, TPW>), dim3((a.M + 15) / 16), dim3(NW * 64), lds, s, a);
    return 0;
}

// [N][K] row-major -> fragment order [N/16][K/EPR][64 lanes][VEC]: lane = (row & 15) + 16 * chunk
template <typename T>
__global__ void pack_weight_frag_kernel(const T* __restrict__ w, T* __restrict__ out, int N, int K) {
    constexpr int VEC = 16 / (int)sizeof(T), EPR = 4 * VEC;
    const int nk = K / EPR;
    const long long total = (long long)(N / 16) * nk * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long tile = i >> 6;
        const int kt = (int)(tile % nk), nt = (int)(tile / nk);
        const int row = nt * 16 + (lane & 15), k = kt * EPR + (lane >> 4) * VEC;
        *reinterpret_cast<uint4*>(out + i * VEC) = *reinterpret_cast<const uint4*>(w + (long long)row * K + k);
    }
}

template <typename T> int dispatch_gl(const GemmLnArgs& a, hipStream_t s) {
    switch (a.H) {
        case 64: return launch_gl<T, 4, 1>(a, s);
        case 128: return launch_gl<T, 8, 1>(a, s);
        case 256: return launch_gl<T, 8, 2>(a, s);
        case 512: return launch_gl<T, 8, 4>(a, s);
        case 1024: return launch_gl<T, 8, 8>(a, s);
        default: mic_set_error("mic_linear_res_ln: hidden size %d has no fused variant", a.H); return MIC_ERR_ARG;
    }
}

}  // namespace

extern "C" int mic_pack_weight_frag(const void* w, void* out, int dtype, int N, int K, void* stream) {
    MIC_ARG_CHECK(w && out && w != out, "mic_pack_weight_frag: bad pointers");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_pack_weight_frag: bad dtype %d", dtype);
    const int epr = 64 / (int)mic_dtype_size(dtype);
    MIC_ARG_CHECK(N > 0 && K > 0 && N % 16 == 0 && K % epr == 0, "mic_pack_weight_frag: N=%d must be a multiple of 16 and K=%d of %d", N, K, epr);
    const long long total = (long long)(N / 16) * (K / epr) * 64;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == MIC_BF16)
        MIC_LAUNCH(pack_weight_frag_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)out, N, K);
    else
        MIC_LAUNCH(pack_weight_frag_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)w, (float*)out, N, K);
    return 0;
}

extern "C" int mic_linear_res_ln_supported(int H) { return H == 64 || H == 128 || H == 256 || H == 512 || H == 1024; }

extern "C" int mic_linear_res_ln(const void* x, int ld_x, const void* w, const float* bias, const float* res, int ld_res,
                                 const float* gamma, const float* beta, float* y, void* yt, int ld_yt, int dtype, int M, int K, int H,
                                 void* stream) {
    MIC_ARG_CHECK(x && w && gamma && beta && (y || yt), "mic_linear_res_ln: null pointer");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_linear_res_ln: bad dtype %d", dtype);
    const int epr = 64 / (int)mic_dtype_size(dtype);
    MIC_ARG_CHECK(M > 0 && K > 0 && K % epr == 0 && ld_x >= K && ld_x % (epr / 4) == 0, "mic_linear_res_ln: bad sizes M=%d K=%d ld_x=%d", M, K, ld_x);
    MIC_ARG_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "mic_linear_res_ln: x and w must be 16-byte aligned");
    MIC_ARG_CHECK(!res || (ld_res >= H && ld_res % 4 == 0), "mic_linear_res_ln: ld_res=%d", ld_res);
    MIC_ARG_CHECK(!yt || (ld_yt >= H && ld_yt % 4 == 0), "mic_linear_res_ln: ld_yt=%d", ld_yt);
    GemmLnArgs a{x, w, bias, res, gamma, beta, y, yt, ld_x, ld_res, ld_yt, M, K, H};
    hipStream_t s = (hipStream_t)stream;
    return dtype == MIC_BF16 ? dispatch_gl<bf16_t>(a, s) : dispatch_gl<float>(a, s);
}
