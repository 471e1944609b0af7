// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   out[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] + res[m, n] )
//
// X is gathered on the fly from a pixel-major (NHWC) activation tensor: row m is output pixel
// (b, oh, ow) and k walks (tap, channel) with `run` contiguous elements per tap, so both operands
// are K-contiguous and feed the MFMA fragments with 16-byte LDS reads.  One k-tile is 64 bytes of
// K per row (32 bf16 or 16 fp32).  The weight fragment is the MFMA A operand and the activation
// fragment the B operand, so every lane ends up with 4 consecutive output channels of one pixel
// and the epilogue stores (and reads bias/residual) 8-16 bytes per lane.
//
// Pipeline per workgroup (256 threads = 2x2 waves): global -> registers for tile k+1 is issued
// before the MFMAs of tile k and written to the other LDS buffer after them; one barrier per tile.
#include "igemm_shared.h"

#include <atomic>
#include <stdlib.h>

#include <mutex>
#include <vector>

namespace {

// With GLDS the LDS holds NS stages and NS-1 tiles are kept in flight: a wave waits (counted vmcnt)
// only for its own pieces of the tile about to be consumed, one barrier publishes the tile and frees
// the buffer consumed in the previous step, which is refilled at once.  Tiles past the end of K are
// requested as zeros so the wait count is the same in every iteration.
//
// A workgroup is WM x WN waves, each owning a (BM/WM) x (BN/WN) sub-tile.  Operand bytes fetched per
// FLOP fall as (BM + BN) / (BM * BN): the kernel is bound by the per-CU fill rate (L2 -> LDS) well
// before the MFMA rate, so the large-M layers take 256-row tiles with 8 waves.
// SEG2: the K range continues into a second input tensor (see IgemmArgs::in2); compiled in only for
// the variants the fused bottleneck tails use, because its extra row pointers cost the plain kernels
// a step of occupancy.
// LN: the tile spans whole output rows (one n-tile) and the epilogue is LayerNorm over the row followed by the optional
// residual add; the fp32 tile is staged 32 rows at a time and every wave then owns whole rows, one float4 per lane per 256
// columns, the summation order of the stand-alone layernorm_kernel (so fusing changes no bit of the result).
// BUFX (register staging only): operands are loaded through buffer descriptors at 32-bit byte offsets; a tap outside the
// image, a row outside the problem and a k past the end of K are bit 31 of the offset, for which the hardware returns zeros.
// One add and two bit operations per load replace two 64-bit multiply-adds, four compares and a branch around the load.
// Needs tensors of less than 2 GiB and at most 31 taps (igemm_fits_buffers).
template <typename T, int BM, int BN, int KT, int WM = 2, int WN = 2, bool RES_PRE = false, bool GLDS = false, int NS = 2, bool SEG2 = false,
          bool LN = false, bool BUFX = false>
__global__ __launch_bounds__(64 * WM * WN, RES_PRE ? 3 : 1) void igemm_kernel(const IgemmArgs a) {
    constexpr int NW = WM * WN, NT = 64 * NW;  // waves, threads
    constexpr int RW = BM / WM, CW = BN / WN;  // rows / columns per wave
    constexpr int VEC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
    constexpr int EPR = 4 * VEC;              // K elements per 64-byte sub-tile row
    constexpr int R4 = 4 * KT;                // 16-byte chunks per row per stage
    constexpr int RSTEP = NT / R4;            // rows covered by one pass of the workgroup's threads
    constexpr int XC = BM / RSTEP, WC = BN / RSTEP;
    constexpr int TM = RW / 16, TN = CW / 16;  // 16x16 MFMA tiles per wave along m / n
    static_assert(BM % RSTEP == 0 && BN % RSTEP == 0 && RW % 16 == 0 && CW % 16 == 0, "tile / workgroup shape");
    constexpr int STAGE = KT * (BM + BN) * 64;  // bytes of one stage: KT sub-tiles of X then of W
    // epilogue staging: fp32 accumulators of PASS_ROWS rows, padded rows (conflict-light b128 writes)
    // (one wave-row of the workgroup at a time when a wave owns 64 rows, else the whole tile at once)
    constexpr int PASS_ROWS = LN ? 32 : RW >= 64 ? RW : BM;
    constexpr int LDC_S = BN + 4;
    static_assert(!LN || (RW % 32 == 0 && !RES_PRE && !SEG2), "LayerNorm epilogue: 32-row passes");
    constexpr int EPI = PASS_ROWS * LDC_S * 4;
    constexpr int SMEM = NS * STAGE > EPI ? NS * STAGE : EPI;
    static_assert(GLDS || NS == 2, "register staging is double-buffered");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // Blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous range of tiles so that
    // the n-tiles reading the same activation rows hit one L2.
    int lid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;

    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ w = (const T*)a.w;

    // Each thread owns one (sub-tile, chunk) column of the stage and XC / WC rows of it, so
    // R4 consecutive lanes read R4*16 contiguous bytes of one row.
    const int within = tid % R4, kt_own = within >> 2, q_own = within & 3, row0 = tid / R4;
    long long x_img[XC];
    int x_h[XC], x_w[XC];
    bool x_ok[XC];
    const T* x2_p[SEG2 ? XC : 1];
#pragma unroll
    for (int i = 0; i < XC; ++i) {
        int m = m0 + row0 + i * RSTEP;
        x_ok[i] = m < a.M;
        m = x_ok[i] ? m : a.M - 1;
        const int b = m / a.OHW, r = m - b * a.OHW, oh = r / a.OW, ow = r - oh * a.OW;
        x_img[i] = (long long)b * a.img_stride + q_own * VEC;
        x_h[i] = oh * a.stride - a.pad_h;
        x_w[i] = ow * a.stride - a.pad_w;
        if constexpr (SEG2)
            x2_p[i] = (const T*)a.in2 + (long long)b * a.img_stride2 + (long long)(oh * a.stride2) * a.row_stride2 +
                      (long long)(ow * a.stride2) * a.pix_stride2 + q_own * VEC;
    }
    const T* w_src = w + (long long)(n0 + row0) * a.Kt + kt_own * EPR + q_own * VEC;
    // weight matrices are padded to a multiple of 128 rows: tiles wider than that must not read past it
    const int n_rows = (a.N + 127) & ~127;
    bool w_ok[WC];
#pragma unroll
    for (int i = 0; i < WC; ++i) w_ok[i] = BN <= 128 || n0 + row0 + i * RSTEP < n_rows;
    const int x_dst = kt_own * BM * 64, w_dst = KT * BM * 64 + kt_own * BN * 64;

    // tap position of this thread's k column (advances by one stage per load)
    int k_cur = kt_own * EPR, kh, kw, c0;
    {
        const int tap = k_cur / a.run;
        c0 = k_cur - tap * a.run;
        kh = tap / a.kw_taps;
        kw = tap - kh * a.kw_taps;
    }
    uint4 xr[XC], wr[WC];
    // buffer form of the same addresses
    unsigned bx_off[BUFX ? XC : 1], bx2_off[BUFX && SEG2 ? XC : 1], bx_bad[BUFX ? XC : 1], bw_off[BUFX ? WC : 1];
    unsigned tap = 0, tapb = 0;  // tap index of k_cur and the byte offset of that tap from tap (0, 0)
    const unsigned rowb = (unsigned)a.row_stride * (unsigned)sizeof(T), pixb = (unsigned)a.pix_stride * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(SEG2 ? a.in2 : a.in), (short)0, (int)0x80000000u, 0x00020000);
    if constexpr (BUFX) {
        const int kh_taps = a.K / a.run / a.kw_taps;
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            bx_off[i] = (unsigned)((x_img[i] + (long long)x_h[i] * a.row_stride + (long long)x_w[i] * a.pix_stride) * (long long)sizeof(T));
            if constexpr (SEG2) bx2_off[i] = (unsigned)((const char*)x2_p[i] - (const char*)a.in2);
            unsigned bad = x_ok[i] ? 0u : 0xffffffffu;
            for (int th = 0, tp = 0; th < kh_taps; ++th)
                for (int tw = 0; tw < a.kw_taps; ++tw, ++tp)
                    if (!((unsigned)(x_h[i] + th) < (unsigned)a.IH && (unsigned)(x_w[i] + tw) < (unsigned)a.IW)) bad |= 1u << tp;
            bx_bad[i] = bad;
        }
#pragma unroll
        for (int i = 0; i < WC; ++i)
            bw_off[i] = (unsigned)(((long long)(n0 + row0 + i * RSTEP) * a.Kt + kt_own * EPR + q_own * VEC) * (long long)sizeof(T)) | (w_ok[i] ? 0u : 0x80000000u);
        tap = (unsigned)(kh * a.kw_taps + kw);
        tapb = (unsigned)kh * rowb + (unsigned)kw * pixb;
    }
    auto load_stage = [&]() {
        if constexpr (BUFX) {
            const unsigned dead = k_cur < a.Kt ? 0u : 0x80000000u;
            const bool seg1 = !SEG2 || k_cur < a.K;
            const unsigned sb = seg1 ? tapb + (unsigned)c0 * (unsigned)sizeof(T) : (unsigned)(k_cur - a.K) * (unsigned)sizeof(T);
            const unsigned bit = seg1 ? tap : 31u;
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                unsigned base = bx_off[i];
                if constexpr (SEG2) base = seg1 ? base : bx2_off[i];
                // an offset that wrapped below zero (a tap above or left of the first image) has bit 31 set as well
                const unsigned off = (base + sb) | (__builtin_amdgcn_ubfe(bx_bad[i], bit, 1) << 31) | dead;
                const auto v = SEG2 ? __builtin_amdgcn_raw_buffer_load_b128(seg1 ? rs_x : rs_x2, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
                xr[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < WC; ++i) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bw_off[i] | dead, 0, 0);
                wr[i] = make_uint4(v[0], v[1], v[2], v[3]);
                bw_off[i] += KT * EPR * (unsigned)sizeof(T);
            }
            k_cur += KT * EPR;
            c0 += KT * EPR;
            while (c0 >= a.run) {
                c0 -= a.run;
                ++tap;
                tapb += pixb;
                if (++kw == a.kw_taps) { kw = 0; ++kh; tapb += rowb - (unsigned)a.kw_taps * pixb; }
            }
            return;
        }
        const bool k_ok = k_cur < a.Kt;
        if (!SEG2 || k_cur < a.K) {
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                const int ih = x_h[i] + kh, iw = x_w[i] + kw;
                const bool ok = k_ok && x_ok[i] && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
                const T* p = in + x_img[i] + (long long)ih * a.row_stride + (long long)iw * a.pix_stride + c0;
                xr[i] = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
            }
        } else if constexpr (SEG2) {  // second segment (or past the end: zeros)
#pragma unroll
            for (int i = 0; i < XC; ++i)
                xr[i] = k_ok && x_ok[i] ? *reinterpret_cast<const uint4*>(x2_p[i] + (k_cur - a.K)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WC; ++i)
            wr[i] = k_ok && w_ok[i] ? *reinterpret_cast<const uint4*>(w_src + (long long)i * RSTEP * a.Kt) : make_uint4(0, 0, 0, 0);
        w_src += KT * EPR;
        k_cur += KT * EPR;
        c0 += KT * EPR;
        while (c0 >= a.run) {
            c0 -= a.run;
            if (++kw == a.kw_taps) { kw = 0; ++kh; }
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XC; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + x_dst + lds_off(row0 + i * RSTEP, q_own)) = xr[i];
#pragma unroll
        for (int i = 0; i < WC; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + w_dst + lds_off(row0 + i * RSTEP, q_own)) = wr[i];
    };

    // ---- direct-to-LDS staging state: instruction j of this wave fills unit u = NW*j + wave ----
    constexpr int NXI = GLDS ? KT * (BM / 16) / NW : 1, NWI = GLDS ? KT * (BN / 16) / NW : 1;
    static_assert(!GLDS || ((KT * (BM / 16)) % NW == 0 && (KT * (BN / 16)) % NW == 0), "units per wave");
    long long gx_img[NXI];
    int gx_h[NXI], gx_w[NXI], gx_k[NXI], gx_c0[NXI], gx_kh[NXI], gx_kw[NXI], gx_lds[NXI];
    bool gx_ok[NXI];
    const T* gx2_p[SEG2 ? NXI : 1];
    const T* gw_src[NWI];
    int gw_k[NWI], gw_lds[NWI];
    bool gw_ok[NWI];
    if constexpr (GLDS) {
        const int lrow = lane >> 2, slot = lane & 3;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int u = NW * j + wave, kt = u / (BM / 16), rg = u % (BM / 16), row = rg * 16 + lrow;
            const int qs = slot ^ (((row >> 3) & 1) << 1);
            int m = m0 + row;
            gx_ok[j] = m < a.M;
            m = gx_ok[j] ? m : a.M - 1;
            const int b = m / a.OHW, r = m - b * a.OHW, oh = r / a.OW, ow = r - oh * a.OW;
            gx_img[j] = (long long)b * a.img_stride + qs * VEC;
            gx_h[j] = oh * a.stride - a.pad_h;
            gx_w[j] = ow * a.stride - a.pad_w;
            gx_k[j] = kt * EPR;
            const int tap = gx_k[j] / a.run;
            gx_c0[j] = gx_k[j] - tap * a.run;
            gx_kh[j] = tap / a.kw_taps;
            gx_kw[j] = tap - gx_kh[j] * a.kw_taps;
            gx_lds[j] = kt * BM * 64 + rg * 1024;
            if constexpr (SEG2)
                gx2_p[j] = (const T*)a.in2 + (long long)b * a.img_stride2 + (long long)(oh * a.stride2) * a.row_stride2 +
                           (long long)(ow * a.stride2) * a.pix_stride2 + qs * VEC;
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            const int u = NW * j + wave, kt = u / (BN / 16), rg = u % (BN / 16), row = rg * 16 + lrow;
            const int qs = slot ^ (((row >> 3) & 1) << 1);
            gw_k[j] = kt * EPR;
            gw_ok[j] = BN <= 128 || n0 + row < n_rows;
            gw_src[j] = w + (long long)(n0 + row) * a.Kt + kt * EPR + qs * VEC;
            gw_lds[j] = KT * BM * 64 + kt * BN * 64 + rg * 1024;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    auto issue_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int ih = gx_h[j] + gx_kh[j], iw = gx_w[j] + gx_kw[j];
            const bool seg1 = gx_k[j] < a.K;
            const bool ok = gx_ok[j] && (seg1 ? (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW : SEG2 && gx_k[j] < a.Kt);
            const void* p = !ok    ? (const void*)g_zero16
                            : seg1 ? (const void*)(in + gx_img[j] + (long long)ih * a.row_stride + (long long)iw * a.pix_stride + gx_c0[j])
                                   : (const void*)(gx2_p[SEG2 ? j : 0] + (gx_k[j] - a.K));
            glds16(p, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * STAGE + gx_lds[j])));
            gx_k[j] += KT * EPR;
            gx_c0[j] += KT * EPR;
#pragma unroll
            for (int t = 0; t < KT; ++t) {  // run >= one 64-byte sub-tile, so at most KT taps are crossed
                const bool wrap = gx_c0[j] >= a.run;
                gx_c0[j] -= wrap ? a.run : 0;
                gx_kw[j] += wrap ? 1 : 0;
                const bool roll = gx_kw[j] == a.kw_taps;
                gx_kw[j] = roll ? 0 : gx_kw[j];
                gx_kh[j] += roll ? 1 : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            const void* p = gw_k[j] < a.Kt && gw_ok[j] ? (const void*)gw_src[j] : (const void*)g_zero16;
            glds16(p, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * STAGE + gw_lds[j])));
            gw_src[j] += KT * EPR;
            gw_k[j] += KT * EPR;
        }
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int xf_off[TM], wf_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xf_off[i] = lds_off(wm * RW + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) wf_off[j] = KT * BM * 64 + lds_off(wn * CW + j * 16 + (lane & 15), lane >> 4);

    constexpr int PASS_ROWS_ = PASS_ROWS, NPASS_ = BM / PASS_ROWS_, CPR_ = BN / 8, CHUNKS_ = PASS_ROWS_ * CPR_ / NT;
    uint4 rpre[RES_PRE ? NPASS_ * CHUNKS_ : 1];
    if constexpr (RES_PRE) {
#pragma unroll
        for (int pass = 0; pass < NPASS_; ++pass)
#pragma unroll
            for (int i = 0; i < CHUNKS_; ++i) {
                const int c = tid + i * NT, r = c / CPR_, ch = c - r * CPR_;
                const int m = m0 + pass * PASS_ROWS_ + r, n = n0 + ch * 8;
                rpre[pass * CHUNKS_ + i] = (m < a.M && n + 7 < a.N)
                                               ? *reinterpret_cast<const uint4*>((const bf16_t*)a.res + (long long)m * a.ld_res + n)
                                               : make_uint4(0, 0, 0, 0);
            }
    }

    auto compute_stage = [&](int buf) {
        const unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            uint4 xf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const uint4*>(base + kt * BM * 64 + xf_off[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(base + kt * BN * 64 + wf_off[j]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
        }
    };
    if constexpr (GLDS) {
        constexpr int G = NXI + NWI;  // direct-to-LDS instructions per wave per stage
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_stage(s);
        int cur = 0, fill = NS - 1;
        for (int st = 0; st < a.nk; ++st) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * G) : "memory");
            issue_stage(fill);
            compute_stage(cur);
            cur = cur + 1 == NS ? 0 : cur + 1;
            fill = fill + 1 == NS ? 0 : fill + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        load_stage();
        store_stage(0);
        __syncthreads();
        int cur = 0;
        for (int st = 0; st < a.nk; ++st) {
            const bool more = st + 1 < a.nk;
            if (more) load_stage();
            compute_stage(cur);
            if (more) store_stage(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // Epilogue through LDS: each lane holds channels n..n+3 of pixel m per 16x16 tile; staging the
    // fp32 tile lets every thread then handle 8 consecutive channels of one pixel, so bias /
    // residual reads and the output store are full-row coalesced 16-byte accesses.
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int NPASS = BM / PASS_ROWS;
    if constexpr (LN) {
        constexpr int NJ = (BN + 255) / 256;     // float4 per lane per row
        constexpr int RPW = PASS_ROWS / NW;      // rows per wave per pass, handled together so their latencies overlap
        constexpr int PPW = RW / PASS_ROWS;      // passes per wave-row
        static_assert(PASS_ROWS % NW == 0, "rows per wave");
        const float fn = (float)a.N;
        bool ok[NJ];
        float4 bias4[NJ], g4[NJ], b4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = lane * 4 + 256 * j;
            ok[j] = n < a.N && n < BN;
            const int nc = ok[j] ? n : 0;
            bias4[j] = a.bias ? *reinterpret_cast<const float4*>(a.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
            g4[j] = *reinterpret_cast<const float4*>(a.ln_g + nc);
            b4[j] = *reinterpret_cast<const float4*>(a.ln_b + nc);
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            // the residual rows of this pass are requested before the tile is staged: their latency hides behind it
            asm volatile("" ::: "memory");  // one pass of residual rows in registers at a time
            float4 rres[RPW][NJ];
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int m = m0 + pass * PASS_ROWS + wave + NW * k;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    rres[k][j] = (a.res && ok[j] && m < a.M)
                                     ? *reinterpret_cast<const float4*>((const float*)a.res + (long long)m * a.ld_res + lane * 4 + 256 * j)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (wm == pass / PPW) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if (i * 16 / PASS_ROWS != pass % PPW) continue;
                        const int r = i * 16 % PASS_ROWS + (lane & 15), cidx = wn * CW + j * 16 + (lane >> 4) * 4;
                        *reinterpret_cast<f32x4_t*>(sC + r * LDC_S + cidx) = acc[j][i];
                    }
            }
            __syncthreads();
            f32x4_t v[RPW][NJ];
            float sum[RPW], mean[RPW], q[RPW], rstd[RPW];
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                sum[k] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    v[k][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (ok[j]) {
                        v[k][j] = *reinterpret_cast<const f32x4_t*>(sC + (wave + NW * k) * LDC_S + lane * 4 + 256 * j);
                        v[k][j][0] += bias4[j].x; v[k][j][1] += bias4[j].y; v[k][j][2] += bias4[j].z; v[k][j][3] += bias4[j].w;
                        sum[k] += ln_sum4(v[k][j][0], v[k][j][1], v[k][j][2], v[k][j][3]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < RPW; ++k) mean[k] = wave_sum(sum[k]) / fn;
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                q[k] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (ok[j]) q[k] += ln_sq4(v[k][j][0], v[k][j][1], v[k][j][2], v[k][j][3], mean[k]);
            }
#pragma unroll
            for (int k = 0; k < RPW; ++k) rstd[k] = 1.0f / sqrtf(wave_sum(q[k]) / fn + 1e-5f);
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int m = m0 + pass * PASS_ROWS + wave + NW * k;
                if (m >= a.M) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (ok[j]) {
                        const int n = lane * 4 + 256 * j;
                        float4 o;
                        o.x = ln_apply(v[k][j][0], mean[k], rstd[k], g4[j].x, b4[j].x) + rres[k][j].x;
                        o.y = ln_apply(v[k][j][1], mean[k], rstd[k], g4[j].y, b4[j].y) + rres[k][j].y;
                        o.z = ln_apply(v[k][j][2], mean[k], rstd[k], g4[j].z, b4[j].z) + rres[k][j].z;
                        o.w = ln_apply(v[k][j][3], mean[k], rstd[k], g4[j].w, b4[j].w) + rres[k][j].w;
                        if (a.out) *reinterpret_cast<float4*>((float*)a.out + (long long)m * a.ldc + n) = o;
                        if (a.ln_yt) {
                            T* t = (T*)a.ln_yt + (long long)m * a.ld_yt + n;
                            t[0] = from_f32<T>(o.x); t[1] = from_f32<T>(o.y); t[2] = from_f32<T>(o.z); t[3] = from_f32<T>(o.w);
                        }
                    }
            }
            if (pass + 1 < NPASS) __syncthreads();
        }
        return;
    }
    constexpr int CPR = BN / 8;                    // 8-channel chunks per row
    constexpr int CHUNKS = PASS_ROWS * CPR / NT;  // per thread per pass
    static_assert(LN || (PASS_ROWS * CPR) % NT == 0, "epilogue chunks per thread");
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (NPASS == 1 || wm == pass) {
            const int rbase = NPASS == 1 ? wm * RW : 0;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = rbase + i * 16 + (lane & 15), cidx = wn * CW + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<f32x4_t*>(sC + r * LDC_S + cidx) = acc[j][i];
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = tid + i * NT, r = c / CPR, ch = c - r * CPR;
            const int m = m0 + pass * PASS_ROWS + r, n = n0 + ch * 8;
            if (m >= a.M || n >= a.N) continue;
            float v[8];
            {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(sC + r * LDC_S + ch * 8);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(sC + r * LDC_S + ch * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            }
            igemm_finish_chunk<RES_PRE>(a, v, m, n, rpre[RES_PRE ? pass * CHUNKS + i : 0]);
        }
        if (pass + 1 < NPASS) __syncthreads();
    }
}

// ---- optional per-launch event timing (mic_profile_begin / mic_profile_end) ----
struct ProfRec {
    hipEvent_t e0, e1;
    int slot, count;  // count: launches between the two events (1, or those a scope gathered)
    double flops;
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
std::atomic<bool> g_prof_on{false};  // read outside the lock by every launch

}  // namespace

namespace {
// A scope brackets a run of back-to-back launches with ONE pair of events: an event between two kernels costs the stream
// about 5 us, more than many of the launches it would time.  Launches inside a scope add their FLOPs and count to it.
thread_local ProfRec* t_scope = nullptr;
thread_local int t_scope_depth = 0;
}  // namespace

void* mic_prof_open(hipStream_t s, int slot, double flops) {
    if (!g_prof_on) return nullptr;
    if (t_scope_depth > 0) {
        if (t_scope) { t_scope->flops += flops; t_scope->count += 1; }
        return nullptr;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    ProfRec* rec = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_on && g_prof_used < g_prof.size()) {
            rec = &g_prof[g_prof_used++];
            rec->slot = slot;
            rec->flops = flops;
            rec->count = 1;
        }
    }
    if (rec && hipEventRecord(rec->e0, s) != hipSuccess) return nullptr;
    return rec;
}

int mic_prof_close(void* rec, hipStream_t s) {
    if (rec) MIC_HIP_CHECK(hipEventRecord(static_cast<ProfRec*>(rec)->e1, s));
    return 0;
}

MicProfScope::MicProfScope(hipStream_t s, int slot) : stream(s) {
    if (t_scope_depth++ > 0) return;  // an enclosing scope already owns the events
    t_scope = nullptr;
    if (void* rec = mic_prof_open_scope(s, slot)) t_scope = static_cast<ProfRec*>(rec);
}

MicProfScope::~MicProfScope() {
    if (--t_scope_depth > 0) return;
    if (t_scope) (void)hipEventRecord(t_scope->e1, stream);
    t_scope = nullptr;
}

void* mic_prof_open_scope(hipStream_t s, int slot) {
    const int depth = t_scope_depth;
    t_scope_depth = 0;  // take a record the ordinary way, then empty it
    ProfRec* rec = static_cast<ProfRec*>(mic_prof_open(s, slot, 0.0));
    t_scope_depth = depth;
    if (rec) rec->count = 0;
    return rec;
}

namespace {

template <typename T, int BM, int BN, int KT, int WM = 2, int WN = 2, bool RES_PRE = false, bool GLDS = false, int NS = 2, bool SEG2 = false,
          bool LN = false, bool BUFX = false>
int launch(IgemmArgs& a, hipStream_t s, double flops) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const int grid = a.tiles_m * a.tiles_n;
    constexpr int kstage = KT * 64 / (int)sizeof(T);
    a.nk = (a.Kt + kstage - 1) / kstage;
    void* rec = mic_prof_open(s, (BM >= 128 ? 0 : 2) + (sizeof(T) == 2 ? 1 : 0), flops);  // tiles of 128 rows and more share a slot
    MIC_LAUNCH((igemm_kernel<T, BM, BN, KT, WM, WN, RES_PRE, GLDS, NS, SEG2, LN, BUFX>), dim3(grid), dim3(64 * WM * WN), 0, s, a);
    return mic_prof_close(rec, s);
}

// the eight-phase 256 x 256 variant (igemm8.hip) exists in bf16 only
template <typename T> int launch8(IgemmArgs& a, hipStream_t s, double flops) {
    if constexpr (sizeof(T) == 2) return mic_igemm8_launch(a, s, flops);
    mic_set_error("mic_conv2d: tile variant 18 is bf16 only");
    return MIC_ERR_ARG;
}

// Tile variants: (BM, BN, KT).  The heuristic below picks one from the problem shape; the
// environment variable MIC_IGEMM_FORCE=<index> pins one for tuning runs.
template <typename T> int dispatch(IgemmArgs& a, hipStream_t s, double flops, int variant) {
    static const int env_forced = [] {
        const char* e = getenv("MIC_IGEMM_FORCE");
        return e ? atoi(e) : -1;
    }();
    if (a.ln_g) {
        // whole rows per tile, so the tile width follows N.  Measured on the Swin-V2-S shapes at batch 512 (M = 1.6 M ... 100 k
        // rows, K = N and 4N): 128-byte stages win for the 128-wide tile, three direct-to-LDS stages for the wider ones.
        if (a.N <= 128) return launch<T, 128, 128, 2, 4, 2, false, false, 2, false, true>(a, s, flops);
        if (a.N <= 256) return launch<T, 128, 256, 1, 2, 4, false, true, 3, false, true>(a, s, flops);
        return launch<T, 128, 384, 1, 2, 4, false, true, 3, false, true>(a, s, flops);
    }
    const int forced = variant >= 0 ? variant : env_forced;
    int v = forced;
    if (v < 0) {
        // measured on MI355X over the ResNet-50 (batch 256) and decode-step shapes: shallow stages win
        // through occupancy unless K is long; N <= 64 wastes half of a 128-wide tile; problems with
        // few tiles take 64x64 tiles, deeper stages when there are fewer of them than CUs
        const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
        if (t256 >= 192 && a.N % 256 == 0 && a.Kt >= 1024) {
            v = 16;  // long K, enough 256x256 tiles for most CUs: half the operand bytes per FLOP
        } else if (sizeof(T) == 2 && t256 >= 150 && t256 <= 256 && a.N % 256 == 0 && a.Kt >= 512) {
            v = 15;  // one 256x256 tile for most CUs and no second round (the cross-K/V projection at batch 256: 23.1 against 25.5 us)
        } else if (t128 >= 192) {
            // fewer 128x128 tiles than CUs: half-width tiles put twice as many CUs to work (the ResNet-50 stage-2 layers at
            // batch 64: 35 against 41 us)
            if (a.N <= 64 || t128 < 256) v = 8;
            else v = a.Kt >= 2048 ? 1 : 0;
        } else {
            const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
            // a long K pays for the deeper stage even with more tiles than CUs (stage-3 3x3 at batch 64: 39 against 52 us)
            v = t64 > 256 && a.Kt < 2048 ? 4 : 5;
        }
    }
    // bf16, long K: the eight-phase 256 x 256 variant where it has a tile for most compute units, or can split K to get there
    // (measured on the ResNet-50 layers at batch 256: stage 2 and 3 layers with K >= 1024)
    bool eight = false;
    if constexpr (sizeof(T) == 2) {
        // (two K segments: from 768 on, the stage-2 tail at 109 against 118 us)
        if (forced < 0 && a.Kt >= (a.in2 ? 768 : 1024) && mic_igemm8_supported(a)) {
            const long long t256 = (long long)((a.M + 255) / 256) * (a.N / 256);
            eight = t256 >= 150 || mic_igemm8_splitk(a) * t256 >= 150;
        }
    }
    if (a.in2) {
        // two-segment K (fused bottleneck tails): measured on the four ResNet-50 shapes at batch 256
        const int epr = 64 / (int)sizeof(T);
        if (forced < 0) v = eight ? 18 : a.N % 256 != 0 ? 0 : (a.Kt < 256 || a.K % (2 * epr) != 0) ? 11 : 16;
        switch (v) {
            case 11: return launch<T, 128, 256, 1, 2, 4, false, false, 2, true>(a, s, flops);
            case 16: return launch<T, 256, 256, 2, 4, 4, false, true, 2, true>(a, s, flops);
            case 18: return launch8<T>(a, s, flops);
            default: return launch<T, 128, 128, 1, 2, 2, false, false, 2, true>(a, s, flops);
        }
    }
    if (eight) v = 18;
    static const int res_pre = [] {
        const char* e = getenv("MIC_IGEMM_RES_PRE");
        return e ? atoi(e) : 1;
    }();
    // residual prefetch variants: bf16 residual with vector-aligned rows, short K (memory-bound layers)
    const bool no_bufx = getenv("MIC_IGEMM_NO_BUFX") != nullptr;  // comparison runs: the pointer form everywhere
    const bool bufx = !no_bufx && !a.in2 && igemm_fits_buffers(a, (int)sizeof(T));
    if (res_pre && a.res && a.res_dtype == MIC_BF16 && a.vec_ok && a.N % 8 == 0 && a.K <= 128 * res_pre) {
        if (v == 0) return launch<T, 128, 128, 1, 2, 2, true>(a, s, flops);
    }
    const int glds = [] {  // read per launch: comparison runs switch it inside one process
        const char* e = getenv("MIC_IGEMM_GLDS");
        return e ? atoi(e) : 0;
    }();
    if (glds == 1) {
        switch (v) {
            case 0: return launch<T, 128, 128, 1, 2, 2, false, true, 4>(a, s, flops);
            case 1: return launch<T, 128, 128, 2, 2, 2, false, true, 3>(a, s, flops);
            case 3: return launch<T, 128, 64, 2, 2, 2, false, true, 3>(a, s, flops);
            case 8: return launch<T, 128, 64, 1, 2, 2, false, true, 4>(a, s, flops);
            default: break;
        }
    } else if (glds == 2) {
        switch (v) {
            case 0: return launch<T, 128, 128, 1, 2, 2, false, true, 3>(a, s, flops);
            case 1: return launch<T, 128, 128, 2, 2, 2, false, true, 2>(a, s, flops);
            case 3: return launch<T, 128, 64, 2, 2, 2, false, true, 2>(a, s, flops);
            case 8: return launch<T, 128, 64, 1, 2, 2, false, true, 3>(a, s, flops);
            default: break;
        }
    } else if (glds == 3) {
        switch (v) {
            case 0: return launch<T, 128, 128, 1, 2, 2, false, true, 6>(a, s, flops);
            case 1: return launch<T, 128, 128, 2, 2, 2, false, true, 4>(a, s, flops);
            case 3: return launch<T, 128, 64, 2, 2, 2, false, true, 4>(a, s, flops);
            case 8: return launch<T, 128, 64, 1, 2, 2, false, true, 6>(a, s, flops);
            default: break;
        }
    }
    // measured on the ResNet-50 layers at batch 256: 3-5 % on the one-sub-tile stages, a loss with two sub-tiles per stage
    // (the tap position is then per thread) and with the residual prefetch
    if (bufx) {
        switch (v) {
            case 0: return launch<T, 128, 128, 1, 2, 2, false, false, 2, false, false, true>(a, s, flops);
            case 8: return launch<T, 128, 64, 1, 2, 2, false, false, 2, false, false, true>(a, s, flops);
            default: break;
        }
    }
    switch (v) {
        case 0: return launch<T, 128, 128, 1>(a, s, flops);
        case 1: return launch<T, 128, 128, 2>(a, s, flops);
        case 2: return launch<T, 128, 128, 4>(a, s, flops);
        case 3: return launch<T, 128, 64, 2>(a, s, flops);
        case 4: return launch<T, 64, 64, 2>(a, s, flops);
        case 5: return launch<T, 64, 64, 4>(a, s, flops);
        case 6: return launch<T, 64, 64, 8>(a, s, flops);
        case 7: return launch<T, 128, 64, 4>(a, s, flops);
        case 8: return launch<T, 128, 64, 1>(a, s, flops);
        case 9: return launch<T, 256, 128, 1, 4, 2>(a, s, flops);
        case 10: return launch<T, 256, 128, 2, 4, 2>(a, s, flops);
        case 11: return launch<T, 128, 256, 1, 2, 4>(a, s, flops);
        case 12: return launch<T, 256, 256, 1, 4, 4>(a, s, flops);
        case 13: return launch<T, 256, 256, 2, 4, 4>(a, s, flops);
        case 14: return launch<T, 256, 128, 1, 4, 2, false, true, 3>(a, s, flops);
        case 15: return launch<T, 256, 256, 1, 4, 4, false, true, 3>(a, s, flops);
        case 16: return launch<T, 256, 256, 2, 4, 4, false, true, 2>(a, s, flops);
        case 17: return launch<T, 256, 64, 1, 4, 1>(a, s, flops);
        case 18: return launch8<T>(a, s, flops);
        default: mic_set_error("mic_conv2d: unknown tile variant %d", v); return MIC_ERR_ARG;
    }
}

}  // namespace

namespace {
constexpr int LN_MAX_N = 384;  // widest tile of the LayerNorm-epilogue variants
}

extern "C" int mic_conv2d_ln_supported(int N) { return N > 0 && N % 4 == 0 && N <= LN_MAX_N; }

namespace {
int conv2d_impl(const mic_conv_desc* d, void* stream, int variant) {
    MIC_ARG_CHECK(d && d->in && d->w && d->out, "mic_conv2d: null pointer");
    MIC_ARG_CHECK(d->dtype == MIC_F32 || d->dtype == MIC_BF16, "mic_conv2d: bad dtype %d", d->dtype);
    MIC_ARG_CHECK(d->out_dtype == MIC_F32 || d->out_dtype == MIC_BF16, "mic_conv2d: bad out_dtype %d", d->out_dtype);
    MIC_ARG_CHECK(!d->res || d->res_dtype == MIC_F32 || d->res_dtype == MIC_BF16, "mic_conv2d: bad res_dtype");
    MIC_ARG_CHECK(d->relu >= 0 && d->relu <= MIC_ACT_GELU, "mic_conv2d: unknown activation %d", d->relu);
    MIC_ARG_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "mic_conv2d: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    const int epr = 64 / (int)mic_dtype_size(d->dtype);
    MIC_ARG_CHECK(d->run > 0 && d->run % epr == 0, "mic_conv2d: run=%d must be a multiple of %d", d->run, epr);
    MIC_ARG_CHECK(d->kw_taps > 0 && d->K % d->run == 0 && (d->K / d->run) % d->kw_taps == 0,
                  "mic_conv2d: K=%d is not taps*run (run=%d, kw_taps=%d)", d->K, d->run, d->kw_taps);
    MIC_ARG_CHECK(d->OH > 0 && d->OW > 0 && d->M % (d->OH * d->OW) == 0, "mic_conv2d: M=%d not a multiple of OH*OW", d->M);
    MIC_ARG_CHECK(d->stride > 0 && d->IH > 0 && d->IW > 0 && d->ldc >= d->N, "mic_conv2d: bad geometry");
    // every tile loads whole 128-row groups of the weight matrix without a per-row guard
    MIC_ARG_CHECK(d->w_rows == 0 || d->w_rows >= (d->N + 127) / 128 * 128,
                  "mic_conv2d: weight matrix has %d rows, N=%d needs it zero-padded to %d", d->w_rows, d->N, (d->N + 127) / 128 * 128);
    const int vec = 16 / (int)mic_dtype_size(d->dtype);
    MIC_ARG_CHECK(d->in_pix_stride % 4 == 0 && d->in_row_stride % vec == 0 && d->in_img_stride % vec == 0 &&
                      ((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->w & 15) == 0,
                  "mic_conv2d: input strides / pointers must keep 16-byte alignment");
    // pixel stride may be smaller than a chunk only when chunks still start 16-byte aligned
    MIC_ARG_CHECK((d->in_pix_stride * (int)mic_dtype_size(d->dtype)) % 8 == 0, "mic_conv2d: pixel stride alignment");
    if (d->in2) {
        MIC_ARG_CHECK(d->K2 > 0 && d->K2 % epr == 0 && d->K % epr == 0, "mic_conv2d: second segment K2=%d (K=%d) must be multiples of %d", d->K2, d->K, epr);
        MIC_ARG_CHECK(d->stride2 > 0 && d->in2_pix_stride % vec == 0 && d->in2_row_stride % vec == 0 && d->in2_img_stride % vec == 0 &&
                          ((uintptr_t)d->in2 & 15) == 0,
                      "mic_conv2d: second input strides / pointer must keep 16-byte alignment");
    }
    if (d->ln_gamma) {
        MIC_ARG_CHECK(mic_conv2d_ln_supported(d->N), "mic_conv2d: LayerNorm epilogue needs N %% 4 == 0 and N <= %d, got %d", LN_MAX_N, d->N);
        MIC_ARG_CHECK(d->ln_beta && d->out_dtype == MIC_F32 && d->relu == MIC_ACT_NONE && !d->in2 && variant < 0,
                      "mic_conv2d: LayerNorm epilogue takes gamma and beta, an fp32 output, no activation, one K segment and no forced variant");
        MIC_ARG_CHECK(!d->res || (d->res_dtype == MIC_F32 && d->ld_res % 4 == 0 && ((uintptr_t)d->res & 15) == 0),
                      "mic_conv2d: LayerNorm epilogue adds an fp32 residual with 16-byte aligned rows");
        MIC_ARG_CHECK(d->ldc % 4 == 0 && ((uintptr_t)d->out & 15) == 0 && (!d->bias || ((uintptr_t)d->bias & 15) == 0) &&
                          ((uintptr_t)d->ln_gamma & 15) == 0 && ((uintptr_t)d->ln_beta & 15) == 0,
                      "mic_conv2d: LayerNorm epilogue needs 16-byte aligned rows and vectors");
        MIC_ARG_CHECK(!d->ln_out || d->ld_ln_out >= d->N, "mic_conv2d: ld_ln_out=%d < N=%d", d->ld_ln_out, d->N);
    }
    IgemmArgs a;
    a.in = d->in; a.w = d->w; a.bias = d->bias; a.res = d->res; a.out = d->out;
    a.ln_g = d->ln_gamma; a.ln_b = d->ln_beta; a.ln_yt = d->ln_out; a.ld_yt = d->ld_ln_out;
    a.scratch = d->scratch; a.scratch_bytes = d->scratch ? d->scratch_bytes : 0; a.splitk = 1;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.in2 = d->in2; a.K2 = d->in2 ? d->K2 : 0; a.Kt = a.K + a.K2;
    a.img_stride2 = d->in2_img_stride; a.row_stride2 = d->in2_row_stride; a.pix_stride2 = d->in2_pix_stride; a.stride2 = d->stride2;
    a.OHW = d->OH * d->OW; a.OW = d->OW; a.IH = d->IH; a.IW = d->IW;
    a.img_stride = d->in_img_stride; a.row_stride = d->in_row_stride;
    a.pix_stride = d->in_pix_stride; a.kw_taps = d->kw_taps; a.run = d->run;
    a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.ldc = d->ldc; a.ld_res = d->ld_res; a.relu = d->relu;
    a.out_dtype = d->out_dtype; a.res_dtype = d->res_dtype;
    // 8 consecutive channels per thread: 16-byte bf16 / 2x16-byte fp32 accesses need these alignments
    a.vec_ok = (d->ldc % 8 == 0) && (((uintptr_t)d->out) % 16 == 0) && (!d->bias || ((uintptr_t)d->bias) % 16 == 0) &&
               (!d->res || (d->ld_res % 8 == 0 && ((uintptr_t)d->res) % 16 == 0));
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * d->M * d->N * ((d->k_valid > 0 ? d->k_valid : d->K) + a.K2);
    return d->dtype == MIC_BF16 ? dispatch<bf16_t>(a, s, flops, variant) : dispatch<float>(a, s, flops, variant);
}
}  // namespace

extern "C" int mic_conv2d(const mic_conv_desc* d, void* stream) { return conv2d_impl(d, stream, -1); }

extern "C" int mic_conv2d_variant(const mic_conv_desc* d, int variant, void* stream) {
    MIC_ARG_CHECK(variant >= 0 && variant <= 18, "mic_conv2d_variant: tile variant %d outside [0,18]", variant);
    return conv2d_impl(d, stream, variant);
}

extern "C" int mic_profile_begin(int max_launches) {
    MIC_ARG_CHECK(max_launches > 0 && max_launches <= (1 << 20), "mic_profile_begin: max_launches=%d", max_launches);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    while (g_prof.size() < (size_t)max_launches) {
        ProfRec r{};
        MIC_HIP_CHECK(hipEventCreate(&r.e0));
        MIC_HIP_CHECK(hipEventCreate(&r.e1));
        g_prof.push_back(r);
    }
    g_prof_used = 0;
    g_prof_on = true;
    return 0;
}

extern "C" int mic_profile_end(int32_t count[4], double ms[4], double flops[4]) {
    MIC_ARG_CHECK(count && ms && flops, "mic_profile_end: null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    for (int i = 0; i < 4; ++i) { count[i] = 0; ms[i] = 0.0; flops[i] = 0.0; }
    for (size_t i = 0; i < g_prof_used; ++i) {
        ProfRec& r = g_prof[i];
        MIC_HIP_CHECK(hipEventSynchronize(r.e1));
        float t = 0.f;
        MIC_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
        const int v = r.slot;  // 2 * (tile variant) + (bf16)
        if (r.count == 0) continue;  // a scope nothing was launched in
        count[v] += r.count; ms[v] += t; flops[v] += r.flops;
    }
    g_prof_used = 0;
    return 0;
}
