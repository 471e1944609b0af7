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
#include "decode_shared.h"

#include <atomic>

#include <stdlib.h>

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
    s = sum_xor16(s);
    s = sum_xor32(s);
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
    sq = sum_xor16(sq);
    sq = sum_xor32(sq);
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

// ---------------------------------------------------------------------------------------------
// Skinny GEMM, general form:  out = act(X W^T + bias + R), same operand streaming, where X and R may
// each be a *LayerNorm view* of a raw fp32 matrix: LN(raw)*gamma+beta evaluated while the rows are
// staged (X) or added (R), from per-row partial statistics the producing launch left behind.  A
// launch can in turn emit those statistics for its own output (mean and centred sum of squares of
// each row over the workgroup's 128 columns).  Chained this way, a post-LN transformer layer needs
// no LayerNorm launch and no launch narrower than (rows/16) x (N/128) workgroups:
//      y, stats = X W^T + b + LN(y_prev)        (producer: plain stores, end of launch)
//      q        = LN(y; stats) Wq^T + bq        (consumer: next launch)
// Partial statistics are combined with the pairwise update of Chan et al., so the result has
// two-pass accuracy.  A workgroup (4 waves) owns MT*16 rows and 128 columns (2 tiles per wave).
struct LnRef {
    const float* raw;
    const float* stats;
    const float *gamma, *beta;
    int ld, parts, width;
};
struct SkinnyArgs {
    const void* x;
    LnRef xl;
    const void* w;
    const float* bias;
    const float* res;
    LnRef rl;
    void* out;
    float* stats_out;
    int ld_x, ld_res, ldc, M, N, K, relu, out_dtype, col_blocks, xcd_rows;
    // LSTM cell in the epilogue (cell_c != NULL): the N = 4 Hd columns are the gate pre-activations in unit-major order
    // (column 4u + g: gate g of hidden unit u, g = input, forget, cell, output), so the four values a lane holds of one
    // 16-column tile are the four gates of one unit.  c is updated in place, h' goes to cell_h (fp32) and, in T, to
    // cell_t0 / cell_t1; nothing is written to `out`.
    float* cell_c;
    float* cell_h;
    void* cell_t0;
    void* cell_t1;
    int cell_ld0, cell_ld1, cell_hd;
};

__device__ __forceinline__ float skinny_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// mean and 1/sqrt(var + eps) of one row from its per-part (mean, M2) pairs.  All parts are requested before any is used
// (a loop with a runtime trip count would wait for each pair in turn: one memory round trip per part); parts past the
// row's own re-read the last one and are skipped in the merge.
constexpr int MAX_LN_PARTS = 8;  // rows of up to 1024 columns
__device__ __forceinline__ void merge_row_stats(const LnRef& l, int m, float& mean, float& rstd) {
    const float2* st = reinterpret_cast<const float2*>(l.stats) + (long long)m * l.parts;
    float2 pr[MAX_LN_PARTS];
#pragma unroll
    for (int p = 0; p < MAX_LN_PARTS; ++p) pr[p] = st[p < l.parts ? p : l.parts - 1];
    float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
    for (int p = 0; p < MAX_LN_PARTS; ++p) {
        if (p < l.parts) chan_step(n, mu, m2, (float)min(128, l.width - 128 * p), pr[p].x, pr[p].y);
    }
    mean = mu;
    rstd = chan_rstd(m2, n);
}

// NWV waves share the 128 columns (8 / NWV tiles of 16 each).  A wave's weight stream is a chain of
// round trips to L2 with 8 fragment loads in flight, so eight waves of one tile each finish it in half
// the round trips of four waves of two.
template <typename T, int MT, int NWV = 4>
// (the second launch bound is waves per SIMD: 4 keeps two 8-wave workgroups on a CU, i.e. at most 128 registers per lane)
__global__ __launch_bounds__(64 * NWV, (MT == 2 && NWV == 4) ? 3 : (MT == 1 && NWV == 8) ? 4 : 1) void skinny_gemm_kernel(const SkinnyArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T), EPR = 4 * VEC, TPW = 8 / NWV, U = (MT >= 4 ? 4 : 8) / TPW, ROWS = MT * 16;
    constexpr int NT = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];
    __shared__ float rs_x[ROWS][2], rs_r[ROWS][2], red[2][NWV][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    // Workgroups b and b+8 run on the same XCD (and share its L2).  All column blocks of a row block go to
    // one XCD, and row block rb always to XCD rb % 8, so that the activation rows are fetched into one L2
    // and the next launch of the chain finds the rows its predecessor wrote on the XCD that wrote them.
    int cb, rb;
    {
        const int rbs = gridDim.x / a.col_blocks;
        if ((rbs & 7) == 0 && a.xcd_rows) {
            const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
            cb = j % a.col_blocks;
            rb = x + 8 * (j / a.col_blocks);
        } else {
            cb = blockIdx.x % a.col_blocks;
            rb = blockIdx.x / a.col_blocks;
        }
    }
    const int m0 = rb * ROWS, nb = cb * 128 + wave * TPW * 16;
    const int nk = a.K / EPR;
    const int xrow = a.K * (int)sizeof(T) + 16;
    // The first 2*U k-tiles of weights (all of K up to 512 bf16 elements) are requested before the
    // activation rows are staged: the two do not depend on each other, so the launch pays one memory
    // round trip for both instead of two in sequence.
    const T* wp = (const T*)a.w + ((long long)(nb / 16) * nk * 64 + lane) * VEC;
    uint4 wa[U][TPW], wb[U][TPW];
    auto load_w = [&](int kt0, uint4 (&wf)[U][TPW]) {
        if (kt0 >= nk) return;  // nothing left to prefetch (wave-uniform)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt0 + u < nk ? kt0 + u : nk - 1;
#pragma unroll
            for (int j = 0; j < TPW; ++j) wf[u][j] = *reinterpret_cast<const uint4*>(wp + ((long long)j * nk + kt) * 64 * VEC);
        }
    };
    load_w(0, wa);
    load_w(U, wb);
    // bias, LayerNorm parameters and residual values of this lane's outputs: also requested up front
    float bv[TPW][4], gv[TPW][4], bev[TPW][4], rv[MT][TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = nb + j * 16 + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = n + e < a.N;
            bv[j][e] = ok && a.bias ? a.bias[n + e] : 0.f;
            gv[j][e] = ok && a.rl.raw ? a.rl.gamma[n + e] : 0.f;
            bev[j][e] = ok && a.rl.raw ? a.rl.beta[n + e] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + i * 16 + r, mc = m < a.M ? m : a.M - 1;
                rv[i][j][e] = !ok ? 0.f : a.res ? a.res[(long long)mc * a.ld_res + n + e] : a.rl.raw ? a.rl.raw[(long long)mc * a.rl.ld + n + e] : 0.f;
            }
        }
    }
    float cprev[MT][TPW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int n = nb + j * 16 + q * 4, m = m0 + i * 16 + r;
            cprev[i][j] = a.cell_c && n < a.N && m < a.M ? a.cell_c[(long long)m * a.cell_hd + (n >> 2)] : 0.f;
        }
    if (tid < ROWS) {
        const int mr = m0 + tid < a.M ? m0 + tid : a.M - 1;
        if (a.xl.raw) merge_row_stats(a.xl, mr, rs_x[tid][0], rs_x[tid][1]);
        if (a.rl.raw) merge_row_stats(a.rl, mr, rs_r[tid][0], rs_r[tid][1]);
    }
    if (a.xl.raw) __syncthreads();
    {
        const int cpr = a.K / VEC;
        for (int c = tid; c < ROWS * cpr; c += NT) {
            const int row = c / cpr, ch = c - row * cpr;
            const int mr = m0 + row < a.M ? m0 + row : a.M - 1;
            if (a.xl.raw) {
                const float mean = rs_x[row][0], rstd = rs_x[row][1];
                const float* src = a.xl.raw + (long long)mr * a.xl.ld + ch * VEC;
                Vec16<T> o;
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src + e4);
                    const float4 g = *reinterpret_cast<const float4*>(a.xl.gamma + ch * VEC + e4);
                    const float4 b = *reinterpret_cast<const float4*>(a.xl.beta + ch * VEC + e4);
                    o.v[e4 + 0] = from_f32<T>(ln_apply(v.x, mean, rstd, g.x, b.x));
                    o.v[e4 + 1] = from_f32<T>(ln_apply(v.y, mean, rstd, g.y, b.y));
                    o.v[e4 + 2] = from_f32<T>(ln_apply(v.z, mean, rstd, g.z, b.z));
                    o.v[e4 + 3] = from_f32<T>(ln_apply(v.w, mean, rstd, g.w, b.w));
                }
                st16(reinterpret_cast<T*>(xs + row * xrow + ch * 16), o);
            } else {
                *reinterpret_cast<uint4*>(xs + row * xrow + ch * 16) = *reinterpret_cast<const uint4*>((const T*)a.x + (long long)mr * a.ld_x + ch * VEC);
            }
        }
    }
    const unsigned char* xp = xs + r * xrow + q * 16;
    __syncthreads();

    f32x4_t acc[MT][TPW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mma_group = [&](int kt0, const uint4 (&wf)[U][TPW]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt0 + u < nk) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint4 xf = *reinterpret_cast<const uint4*>(xp + i * 16 * xrow + (kt0 + u) * 64);
#pragma unroll
                    for (int j = 0; j < TPW; ++j) SkinnyMma<T>::run(wf[u][j], xf, acc[i][j]);
                }
            }
        }
    };
    for (int kt = 0; kt < nk; kt += 2 * U) {
        mma_group(kt, wa);
        load_w(kt + 2 * U, wa);
        mma_group(kt + U, wb);
        load_w(kt + 3 * U, wb);
    }

    // epilogue values: lane holds columns nb + j*16 + q*4 + e of rows m0 + i*16 + r
    float v[MT][TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = nb + j * 16 + q * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[i][j][e] + bv[j][e];
                if (n + e < a.N) {
                    if (a.res) x += rv[i][j][e];
                    if (a.rl.raw) x += ln_apply(rv[i][j][e], rs_r[i * 16 + r][0], rs_r[i * 16 + r][1], gv[j][e], bev[j][e]);
                } else {
                    x = 0.f;
                }
                v[i][j][e] = a.relu ? fmaxf(x, 0.f) : x;
            }
        }
    }
    if (a.cell_c) {
        // the arithmetic of lstm_cell_kernel (decode_ops.hip) on the values that kernel would have read back from memory
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int n = nb + j * 16 + q * 4, u = n >> 2;
            if (n >= a.N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + i * 16 + r;
                if (m >= a.M) continue;
                const float ig = skinny_sigmoid(v[i][j][0]), fg = skinny_sigmoid(v[i][j][1]), gg = tanhf(v[i][j][2]), og = skinny_sigmoid(v[i][j][3]);
                const float cn = __builtin_fmaf(fg, cprev[i][j], ig * gg);
                const float hn = og * tanhf(cn);
                const long long ci = (long long)m * a.cell_hd + u;
                a.cell_c[ci] = cn;
                a.cell_h[ci] = hn;
                if (a.cell_t0) ((T*)a.cell_t0)[(long long)m * a.cell_ld0 + u] = from_f32<T>(hn);
                if (a.cell_t1) ((T*)a.cell_t1)[(long long)m * a.cell_ld1 + u] = from_f32<T>(hn);
            }
        }
        return;
    }
    if (a.stats_out) {
        // per row: mean and centred sum of squares over this workgroup's valid columns
        const float cnt = (float)min(128, a.N - cb * 128);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) s += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
            s = sum_xor16(s);
            s = sum_xor32(s);
            if (q == 0) red[0][wave][i * 16 + r] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = i * 16 + r;
            float rsum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NWV; ++w8) rsum += red[0][w8][row];
            const float mean = rsum / cnt;
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int n = nb + j * 16 + q * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = n + e < a.N ? v[i][j][e] - mean : 0.f;
                    sq = fmaf(d, d, sq);
                }
            }
            sq = sum_xor16(sq);
            sq = sum_xor32(sq);
            if (q == 0) red[1][wave][row] = sq;
        }
        __syncthreads();
        if (tid < ROWS && m0 + tid < a.M) {
            float rsum = 0.f, m2 = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NWV; ++w8) { rsum += red[0][w8][tid]; m2 += red[1][w8][tid]; }
            const float mean = rsum / cnt;
            float* dst = a.stats_out + ((long long)(m0 + tid) * a.col_blocks + cb) * 2;
            dst[0] = mean;
            dst[1] = m2;
        }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = nb + j * 16 + q * 4;
        if (n >= a.N) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + i * 16 + r;
            if (m >= a.M) continue;
            const bool full = n + 3 < a.N && (a.ldc & 3) == 0;
            if (a.out_dtype == MIC_F32) {
                float* o = (float*)a.out + (long long)m * a.ldc + n;
                if (full) *reinterpret_cast<float4*>(o) = make_float4(v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
                else
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) o[e] = v[i][j][e];
            } else {
                bf16_t* o = (bf16_t*)a.out + (long long)m * a.ldc + n;
                if (full) {
                    struct alignas(8) B4 { bf16_t v[4]; } p;
#pragma unroll
                    for (int e = 0; e < 4; ++e) p.v[e] = (bf16_t)v[i][j][e];
                    *reinterpret_cast<B4*>(o) = p;
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) o[e] = (bf16_t)v[i][j][e];
                }
            }
        }
    }
}

template <typename T, int MT, int NWV = 4> int launch_skinny(SkinnyArgs& a, hipStream_t s) {
    const size_t lds = (size_t)MT * 16 * ((size_t)a.K * sizeof(T) + 16);
    MIC_ARG_CHECK(lds <= 140 * 1024, "mic_linear_skinny: K=%d needs %zu bytes of LDS", a.K, lds);
    if (lds > 48 * 1024) {
        static std::atomic<uint64_t> granted{0};  // one bit per device; lanes call from several host threads
        MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&skinny_gemm_kernel<T, MT, NWV>), 140 * 1024, granted));
    }
    a.col_blocks = (a.N + 127) / 128;
    static const int xcd_rows = [] { const char* e = getenv("MIC_SKINNY_XCD_ROWS"); return e ? atoi(e) : 1; }();
    a.xcd_rows = xcd_rows;
    const int grid = a.col_blocks * ((a.M + MT * 16 - 1) / (MT * 16));
    MIC_LAUNCH((skinny_gemm_kernel<T, MT, NWV>), dim3(grid), dim3(64 * NWV), lds, s, a);
    return 0;
}

int check_ln_ref(const char* who, const mic_ln_ref& l, int width) {
    if (!l.raw) return 0;
    MIC_ARG_CHECK(l.stats && l.gamma && l.beta, "%s: LayerNorm view without stats/gamma/beta", who);
    MIC_ARG_CHECK(l.width == width && l.width % 4 == 0 && l.ld >= l.width && l.ld % 4 == 0 && l.parts == (l.width + 127) / 128,
                  "%s: LayerNorm view of width %d (ld %d, %d parts) does not match %d columns", who, l.width, l.ld, l.parts, width);
    MIC_ARG_CHECK(l.parts <= MAX_LN_PARTS, "%s: LayerNorm view of %d columns (at most %d)", who, l.width, 128 * MAX_LN_PARTS);
    MIC_ARG_CHECK(((uintptr_t)l.raw & 15) == 0 && ((uintptr_t)l.gamma & 15) == 0 && ((uintptr_t)l.beta & 15) == 0 && ((uintptr_t)l.stats & 7) == 0,
                  "%s: 16-byte alignment (statistics: 8-byte)", who);
    return 0;
}

template <typename T, int NW, int TPW> int launch_gl(const GemmLnArgs& a, hipStream_t s) {
    const size_t lds = 16 * ((size_t)a.K * sizeof(T) + 16);
    MIC_ARG_CHECK(lds <= 150 * 1024, "mic_linear_res_ln: K=%d needs %zu bytes of LDS", a.K, lds);
    if (lds > 48 * 1024) {
        static std::atomic<uint64_t> granted{0};  // one bit per device
        MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&gemm_res_ln_kernel<T, NW, TPW>), 150 * 1024, granted));
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
        case 512: return launch_gl<T, 16, 2>(a, s);  // 16 waves: twice the loads in flight per workgroup
        case 1024: return launch_gl<T, 16, 4>(a, s);
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

namespace {
struct CellOut {
    float* c;
    float* h;
    void* t0;
    void* t1;
    int ld0, ld1, hd;
};
int skinny_impl(const mic_skinny_desc* d, const CellOut* cell, void* stream);
}  // namespace

extern "C" int mic_linear_skinny_ex(const mic_skinny_desc* d, void* stream) { return skinny_impl(d, nullptr, stream); }

extern "C" int mic_lstm_gates_cell(const mic_skinny_desc* d, float* c, float* h, void* ht0, int ld0, void* ht1, int ld1, int Hd, void* stream) {
    MIC_ARG_CHECK(d && c && h && Hd > 0 && d->N == 4 * Hd, "mic_lstm_gates_cell: c, h and N = 4 Hd columns are needed (N=%d, Hd=%d)", d ? d->N : 0, Hd);
    MIC_ARG_CHECK(!d->relu && !d->res && !d->res_ln.raw && !d->stats_out, "mic_lstm_gates_cell: no activation, residual or statistics");
    MIC_ARG_CHECK((!ht0 || ld0 >= Hd) && (!ht1 || ld1 >= Hd), "mic_lstm_gates_cell: ld0=%d / ld1=%d shorter than Hd=%d", ld0, ld1, Hd);
    // the activation rows are read by every workgroup of the launch: h' must not land in them
    const size_t esz = mic_dtype_size(d->dtype);
    const char *x0 = (const char*)d->x, *x1 = x0 ? x0 + ((size_t)(d->M - 1) * d->ld_x + d->K) * esz : nullptr;
    auto inside = [&](const void* p, int ld) { return p && x0 && (const char*)p + ((size_t)(d->M - 1) * ld + Hd) * esz > x0 && (const char*)p < x1; };
    MIC_ARG_CHECK(!inside(ht0, ld0) && !inside(ht1, ld1), "mic_lstm_gates_cell: h' would overwrite the rows the launch is reading");
    const CellOut cell{c, h, ht0, ht1, ld0, ld1, Hd};
    return skinny_impl(d, &cell, stream);
}

namespace {
int skinny_impl(const mic_skinny_desc* d, const CellOut* cell, void* stream) {
    MIC_ARG_CHECK(d && d->wf && (d->out || cell) && (d->x || d->x_ln.raw), "mic_linear_skinny: null pointer");
    MIC_ARG_CHECK((d->dtype == MIC_F32 || d->dtype == MIC_BF16) && (d->out_dtype == MIC_F32 || d->out_dtype == MIC_BF16),
                  "mic_linear_skinny: bad dtype");
    const int epr = 64 / (int)mic_dtype_size(d->dtype);
    MIC_ARG_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->K % epr == 0, "mic_linear_skinny: bad sizes M=%d N=%d K=%d", d->M, d->N, d->K);
    MIC_ARG_CHECK(d->x_ln.raw || (d->ld_x >= d->K && d->ld_x % (epr / 4) == 0 && ((uintptr_t)d->x & 15) == 0), "mic_linear_skinny: bad x / ld_x=%d", d->ld_x);
    MIC_ARG_CHECK(((uintptr_t)d->wf & 15) == 0 && ((uintptr_t)d->out & 15) == 0, "mic_linear_skinny: 16-byte alignment");
    MIC_ARG_CHECK(cell || d->ldc >= d->N, "mic_linear_skinny: ldc=%d < N=%d", d->ldc, d->N);
    MIC_ARG_CHECK(!d->res || d->ld_res >= d->N, "mic_linear_skinny: ld_res=%d", d->ld_res);
    MIC_ARG_CHECK(!(d->stats_out && d->relu), "mic_linear_skinny: statistics of a ReLU output are not supported");
    MIC_TRY(check_ln_ref("mic_linear_skinny(x_ln)", d->x_ln, d->K));
    MIC_TRY(check_ln_ref("mic_linear_skinny(res_ln)", d->res_ln, d->N));
    SkinnyArgs a{};
    a.x = d->x; a.w = d->wf; a.bias = d->bias; a.res = d->res; a.out = d->out; a.stats_out = d->stats_out;
    a.xl = LnRef{d->x_ln.raw, d->x_ln.stats, d->x_ln.gamma, d->x_ln.beta, d->x_ln.ld, d->x_ln.parts, d->x_ln.width};
    a.rl = LnRef{d->res_ln.raw, d->res_ln.stats, d->res_ln.gamma, d->res_ln.beta, d->res_ln.ld, d->res_ln.parts, d->res_ln.width};
    a.ld_x = d->ld_x; a.ld_res = d->ld_res; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.K = d->K; a.relu = d->relu; a.out_dtype = d->out_dtype;
    if (cell) {
        a.cell_c = cell->c; a.cell_h = cell->h; a.cell_t0 = cell->t0; a.cell_t1 = cell->t1;
        a.cell_ld0 = cell->ld0; a.cell_ld1 = cell->ld1; a.cell_hd = cell->hd;
    }
    hipStream_t s = (hipStream_t)stream;
    // every row block re-reads its 128 weight columns from L2: 32-row blocks halve that traffic and pay
    // off once there are still two workgroups per CU (the vocabulary projection)
    static const int forced_mt = [] { const char* e = getenv("MIC_SKINNY_MT"); return e ? atoi(e) : 0; }();
    const long long cbs = (d->N + 127) / 128;
    static const int mt2_min = [] { const char* e = getenv("MIC_SKINNY_MT2_MIN"); return e ? atoi(e) : 512; }();
    int mt = cbs * ((d->M + 31) / 32) >= mt2_min ? 2 : 1;  // measured: 64-row blocks never win at these sizes
    if (forced_mt) mt = forced_mt;
    static const int forced_nw = [] { const char* e = getenv("MIC_SKINNY_NW"); return e ? atoi(e) : 0; }();
    // measured inside the captured decode step: eight waves win for 16-row blocks; the 32-row blocks of
    // the vocabulary projection need their three workgroups per CU and keep four
    int nw = mt == 1 ? 8 : 4;
    if (forced_nw) nw = forced_nw;
    if (nw == 8 && mt <= 2) {
        if (mt == 2) return d->dtype == MIC_BF16 ? launch_skinny<bf16_t, 2, 8>(a, s) : launch_skinny<float, 2, 8>(a, s);
        return d->dtype == MIC_BF16 ? launch_skinny<bf16_t, 1, 8>(a, s) : launch_skinny<float, 1, 8>(a, s);
    }
    if (mt == 4) return d->dtype == MIC_BF16 ? launch_skinny<bf16_t, 4>(a, s) : launch_skinny<float, 4>(a, s);
    if (mt == 2) return d->dtype == MIC_BF16 ? launch_skinny<bf16_t, 2>(a, s) : launch_skinny<float, 2>(a, s);
    return d->dtype == MIC_BF16 ? launch_skinny<bf16_t, 1>(a, s) : launch_skinny<float, 1>(a, s);
}
}  // namespace

extern "C" int mic_linear_skinny(const void* x, int ld_x, const void* wf, const float* bias, void* out, int out_dtype, int ldc,
                                 int dtype, int M, int N, int K, int relu, void* stream) {
    mic_skinny_desc d{};
    d.x = x; d.ld_x = ld_x; d.wf = wf; d.bias = bias; d.out = out; d.out_dtype = out_dtype; d.ldc = ldc;
    d.dtype = dtype; d.M = M; d.N = N; d.K = K; d.relu = relu;
    return mic_linear_skinny_ex(&d, stream);
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
