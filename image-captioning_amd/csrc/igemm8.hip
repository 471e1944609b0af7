// Implicit-GEMM convolution / linear layer, bf16, 256 x 256 output tile per workgroup of 8 waves, for the layers with a
// long K and enough tiles to fill the chip (same arithmetic and epilogue as igemm.hip; see there for the operand layout).
//
// Waves are 2 (pixel rows) x 4 (channels): a wave owns 128 pixels x 64 channels = 8 x 4 MFMA tiles, 128 accumulator
// registers.  K advances in tiles of 64 elements.  One K-tile of operands is four half-tiles of 16 KiB in LDS:
//   X0 = pixel rows {0..63} of both wave rows, X1 = rows {64..127} of both, W0 = channels {0..31} of the four wave
//   columns, W1 = channels {32..63}: every wave finds the operands of one quadrant of its output in one X half and one
//   W half.  A half-tile is 128 rows x 128 bytes, the 16-byte chunks of a row permuted by an XOR swizzle.
// Two K-tiles are resident (128 KiB).  A K-tile is computed in four phases, one output quadrant each:
//   phase 1  read W0, X0    request X1 of tile t+1    quadrant (0,0)
//   phase 2  read W1        request W0 of tile t+2    quadrant (0,1)
//   phase 3  read X1        request X0 of tile t+2    quadrant (1,1)
//   phase 4                 request W1 of tile t+2    quadrant (1,0)   wait until tile t+1 has landed
// Every phase is  { LDS reads, direct-to-LDS requests } barrier { 16 MFMAs } barrier.  The two wave rows run half a
// phase apart (the second executes one barrier more before the loop, the first one more after it): a SIMD holds one wave
// of each, so while one issues MFMAs the other reads LDS and issues requests.  A half-tile is requested again only after
// a barrier that every wave reaches after its reads of that half-tile have returned (W0: the counted wait before the
// first barrier of phase 1; the others: two phases later), and read only after a barrier that every wave reaches after
// the counted wait that retires its own requests (one barrier more than that for the leading wave row).  Three
// half-tiles stay in flight across every barrier; the vector-memory counter never drains inside the loop.
#include "igemm_shared.h"

#include <stdlib.h>

namespace {

constexpr int HT = 16384;      // bytes of one half-tile
constexpr int KBUF = 4 * HT;   // one K-tile: X0 | W0 | W1 | X1
constexpr int OFF_X0 = 0, OFF_W0 = HT, OFF_W1 = 2 * HT, OFF_X1 = 3 * HT;

#define MIC_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// Buffer descriptor of 2 GiB starting at p (raw 32-bit format): loads through it at byte offsets of 2^31 and more return
// zeros instead of touching memory.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u_t make_srd(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    return v4u_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
}
// One buffer_load_dwordx4 ... lds: every lane fetches 16 bytes at its own byte offset into the buffer, the wave's 1 KiB
// lands at LDS address `lds_base` + 16 * lane (see glds16 for why this is inline assembly and how M0 is handled).
__device__ __forceinline__ void blds16(unsigned voff, const v4u_t& srd, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_base), "s"(srd)
                 : "memory");
}

// LINES: a direct-to-LDS piece (one wave-instruction, 1 KiB) is 8 rows x 128 bytes, whole cache lines, and the image has
// rows of 128 bytes with chunk c of row r at chunk c ^ ((r >> 1) & 7); otherwise a piece is 16 rows x 64 bytes and the image
// is two planes of 64-byte rows in the layout of lds_off().  Both images serve the 16 lanes of an LDS access group from 16
// different slots of a 256-byte bank row.
// DBG (measurement builds only): 1 = no operand requests inside the loop and 2 = every request reads the 16-byte block of
// zeros (both give wrong results; 5 / 6 = only the X / only the W requests do; 7 = requests only, no LDS reads and no MFMAs; 8 = only output row 0 is stored); 3 = wave 0 of every workgroup stamps the cycle counter and the 100 MHz counter around
// the loop, and the latter at kernel entry and after its last store, into the buffer passed as ln_yt (8 x 64-bit words per
// workgroup)
// XS3: three K-tiles of X and two of W are resident (160 KiB): X requests run eight to nine phases ahead of their
// use instead of six (the gathered activation rows are the requests that miss the L2), four half-tiles stay in flight.
// BUFL: operands are requested through buffer descriptors at 32-bit byte offsets (one address register per lane, a tap
// outside the image or a row outside the problem is bit 31 of the offset: the hardware returns zeros), which takes a
// fraction of the address arithmetic of the 64-bit pointer form; needs tensors of less than 2 GiB and at most 31 taps.
template <bool SEG2, int DBG = 0, bool LINES = false, bool XS3 = false, bool BUFL = false> __global__ __launch_bounds__(512) void igemm8_kernel(const IgemmArgs a) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int LDC_S = BN + 4;
    constexpr int EPI = 128 * LDC_S * 4;  // fp32 staging of one wave row's 128 pixel rows
    constexpr int SMEM = XS3 ? 10 * HT : EPI > 2 * KBUF ? EPI : 2 * KBUF;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    unsigned long long tr_e0 = 0;
    if constexpr (DBG == 3) tr_e0 = __builtin_amdgcn_s_memrealtime();

    // blocks b and b+8 share an XCD (and its L2): every XCD takes a contiguous range of tiles
    int lid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;

    // ---- staging: wave w fills rows 16w..16w+15 of every half-tile with two pieces ----
    constexpr int NR = LINES ? 2 : 1;  // rows of a half-tile a thread fetches from
    const bf16_t* __restrict__ in = (const bf16_t*)a.in;
    const bf16_t* xp[2][NR];
    const bf16_t* x2p[2][NR];
    int xh[2][NR], xw[2][NR];
    bool xok[2][NR];
    const bf16_t* wp[2][NR];
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        const int lr = LINES ? wave * 16 + ri * 8 + (lane >> 3) : wave * 16 + (lane >> 2);
        const int ch = LINES ? (lane & 7) ^ ((lr >> 1) & 7) : (lane & 3) ^ (((lr >> 3) & 1) << 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
            xok[h][ri] = m < a.M;
            m = xok[h][ri] ? m : a.M - 1;
            const int b = m / a.OHW, r = m - b * a.OHW, oh = r / a.OW, ow = r - oh * a.OW;
            xh[h][ri] = oh * a.stride - a.pad_h;
            xw[h][ri] = ow * a.stride - a.pad_w;
            xp[h][ri] = in + (long long)b * a.img_stride + (long long)xh[h][ri] * a.row_stride + (long long)xw[h][ri] * a.pix_stride + ch * 8;
            x2p[h][ri] = xp[h][ri];
            if constexpr (SEG2)
                x2p[h][ri] = (const bf16_t*)a.in2 + (long long)b * a.img_stride2 + (long long)(oh * a.stride2) * a.row_stride2 +
                             (long long)(ow * a.stride2) * a.pix_stride2 + ch * 8;
            wp[h][ri] = (const bf16_t*)a.w + (long long)(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31)) * a.Kt + ch * 8;
        }
    }

    // buffer form: byte offsets of the same rows; bit i of xbad = tap i of the row lies outside the image, bit 31 = the row
    // lies outside the problem
    unsigned xoff[2][NR], xbad[2][NR], woff[2][NR];
    const v4u_t srd_x = make_srd(a.in), srd_w = make_srd(a.w), srd_x2 = make_srd(SEG2 ? a.in2 : a.in);
    if constexpr (BUFL) {
        const int kh_taps = a.K / a.run / a.kw_taps;
#pragma unroll
        for (int ri = 0; ri < NR; ++ri)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xoff[h][ri] = (unsigned)((const char*)xp[h][ri] - (const char*)a.in);
                woff[h][ri] = (unsigned)((const char*)wp[h][ri] - (const char*)a.w);
                unsigned bad = xok[h][ri] ? 0u : 0xffffffffu;
                for (int kh = 0, tp = 0; kh < kh_taps; ++kh)
                    for (int kw = 0; kw < a.kw_taps; ++kw, ++tp)
                        if (!((unsigned)(xh[h][ri] + kh) < (unsigned)a.IH && (unsigned)(xw[h][ri] + kw) < (unsigned)a.IW)) bad |= 1u << tp;
                xbad[h][ri] = bad;
                if constexpr (SEG2) xp[h][ri] = (const bf16_t*)(uintptr_t)(unsigned)((const char*)x2p[h][ri] - (const char*)a.in2);
            }
    }

    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned piece = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * (LINES ? 2048u : 1024u));
    constexpr unsigned PIECE2 = LINES ? 1024u : 8192u;  // LDS offset of a wave's second piece of a half-tile
    constexpr int SRC2 = LINES ? 0 : 32;                // and its source offset in elements (the second 64-byte half of the rows)

    // K position of the X requests (the same for every thread): K-tile index, tap, channel offset inside the tap
    int s_t = 0, s_c0 = 0, s_kh = 0, s_kw = 0, s_tap = 0;
    long long s_off = 0;
    auto advance = [&]() {
        ++s_t;
        s_c0 += BK;
        if (s_c0 >= a.run) {
            s_c0 = 0;
            ++s_tap;
            if (++s_kw == a.kw_taps) { s_kw = 0; ++s_kh; }
        }
        s_off = (long long)s_kh * a.row_stride + (long long)s_kw * a.pix_stride + s_c0;
    };
    auto stage_x = [&](int h, unsigned dst) {
        const int k = s_t * BK;
        const bool seg1 = k < a.K;
        if constexpr (DBG == 1) { if (s_t > 1) return; }
        if constexpr (BUFL) {
            // K-tiles past the end of K are requested as zeros like rows outside the problem (xbad bit 31)
            const unsigned sb = (unsigned)(seg1 ? (int)s_off : k - a.K) * 2u;
            const unsigned dead = k < a.Kt ? 0u : 0x80000000u;
            const int bit = seg1 ? s_tap : 31;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int ri = LINES ? pc : 0;
                unsigned base = xoff[h][ri];
                if constexpr (SEG2) base = seg1 ? base : (unsigned)(uintptr_t)xp[h][ri];
                const unsigned bad = __builtin_amdgcn_ubfe(xbad[h][ri], bit, 1);
                const unsigned off = (base + sb + pc * SRC2 * 2u) | (bad << 31) | dead;  // an offset that wrapped below zero has bit 31 set as well
                if constexpr (SEG2) blds16(off, seg1 ? srd_x : srd_x2, piece + dst + pc * PIECE2);
                else blds16(off, srd_x, piece + dst + pc * PIECE2);
            }
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int ri = LINES ? pc : 0;
            bool ok = xok[h][ri] && (seg1 ? (unsigned)(xh[h][ri] + s_kh) < (unsigned)a.IH && (unsigned)(xw[h][ri] + s_kw) < (unsigned)a.IW
                                          : SEG2 && k < a.Kt);
            const bf16_t* p = xp[h][ri] + s_off + pc * SRC2;
            if constexpr (SEG2) p = seg1 ? p : x2p[h][ri] + (k - a.K) + pc * SRC2;
            if constexpr (DBG == 2 || DBG == 5) ok = ok && s_t <= 1;
            glds16(ok ? (const void*)p : (const void*)g_zero16, piece + dst + pc * PIECE2);
        }
    };
    int w_t[2] = {0, 0};
    auto stage_w = [&](int h, unsigned dst) {
        bool ok = w_t[h] * BK < a.Kt;
        if constexpr (DBG == 1) { if (w_t[h] > 1) { ++w_t[h]; return; } }
        if constexpr (DBG == 2 || DBG == 6) ok = ok && w_t[h] <= 1;
        if constexpr (BUFL) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int ri = LINES ? pc : 0;
                blds16((woff[h][ri] + pc * SRC2 * 2u) | (ok ? 0u : 0x80000000u), srd_w, piece + dst + pc * PIECE2);
            }
#pragma unroll
            for (int ri = 0; ri < NR; ++ri) woff[h][ri] += BK * 2;
            ++w_t[h];
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int ri = LINES ? pc : 0;
            glds16(ok ? (const void*)(wp[h][ri] + pc * SRC2) : (const void*)g_zero16, piece + dst + pc * PIECE2);
        }
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) wp[h][ri] += BK;
        ++w_t[h];
    };

    // ---- fragment reads: lane l holds row (l & 15) of a 16-row tile and the 16-byte chunk (l >> 4) of each 64-byte half
    // of the row's 128 bytes ----
    const unsigned fsw = ((lane & 15) >> 1) & 7;
    const unsigned xrd0 = LINES ? (unsigned)(wm * 64 + (lane & 15)) * 128u + ((((unsigned)lane >> 4) ^ fsw) << 4)
                                : (unsigned)lds_off(wm * 64 + (lane & 15), lane >> 4);
    const unsigned wrd0 = LINES ? (unsigned)(wn * 32 + (lane & 15)) * 128u + ((((unsigned)lane >> 4) ^ fsw) << 4)
                                : (unsigned)lds_off(wn * 32 + (lane & 15), lane >> 4);
    const unsigned xrd1 = LINES ? xrd0 ^ 64u : xrd0 + 8192u, wrd1 = LINES ? wrd0 ^ 64u : wrd0 + 8192u;
    constexpr int RT = LINES ? 2048 : 1024;  // bytes between 16-row tiles
    uint4 xa[4][2] = {}, wb0[2][2] = {}, wb1[2][2] = {};
    auto ldx = [&](unsigned off) {
        if constexpr (DBG == 7) return;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i][kp] = *reinterpret_cast<const uint4*>(smem + off + i * RT + (kp ? xrd1 : xrd0));
    };
    auto ldw = [&](uint4 (&wb)[2][2], unsigned off) {
        if constexpr (DBG == 7) return;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int j = 0; j < 2; ++j) wb[j][kp] = *reinterpret_cast<const uint4*>(smem + off + j * RT + (kp ? wrd1 : wrd0));
    };

    unsigned long long tr_c0 = 0, tr_r0 = 0;

    f32x4_t acc[4][8];  // [channel tile][pixel-row tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto quadrant = [&](const uint4 (&wb)[2][2], int hx, int hw) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (DBG != 7)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) Mma<bf16_t>::run(wb[jj][kp], xa[ii][kp], acc[hw * 2 + jj][hx * 4 + ii]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        MIC_RAW_BARRIER();
    };

    // one K-tile: operands read at x0 / x1 / w0 / w1 (LDS offsets of the half-tiles); the W halves and X0 are requested
    // again into the places they were read from, the X1 request of phase 1 goes to x1_dst
    auto tile = [&](unsigned x0, unsigned x1, unsigned w0, unsigned w1, unsigned x1_dst) {
        // phase 1
        ldw(wb0, w0);
        __builtin_amdgcn_sched_barrier(0);
        ldx(x0);
        stage_x(1, x1_dst);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the W0 reads, issued first, have returned
        MIC_RAW_BARRIER();
        quadrant(wb0, 0, 0);
        // phase 2
        ldw(wb1, w1);
        stage_w(0, w0);
        __builtin_amdgcn_sched_barrier(0);
        MIC_RAW_BARRIER();
        quadrant(wb1, 0, 1);
        // phase 3
        ldx(x1);
        advance();
        stage_x(0, x0);
        __builtin_amdgcn_sched_barrier(0);
        MIC_RAW_BARRIER();
        quadrant(wb1, 1, 1);
        // phase 4: wait for all but the half-tiles requested in this K-tile's phases (XS3: four; else the last three):
        // tile t+1 has landed
        stage_w(1, w1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 1) {
            if constexpr (XS3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        MIC_RAW_BARRIER();
        quadrant(wb0, 1, 0);
    };

    if constexpr (XS3) {
        // X slot s at 2 s HT (X0, X1), W slot s at (6 + 2 s) HT (W0, W1).  Prologue: tile 0, tile 1, X0 of tile 2
        constexpr unsigned WB = 6 * HT;
        stage_x(0, 0);
        stage_w(0, WB);
        stage_w(1, WB + HT);
        stage_x(1, HT);
        advance();
        stage_x(0, 2 * HT);
        stage_x(1, 3 * HT);
        stage_w(0, WB + 2 * HT);
        advance();
        stage_x(0, 4 * HT);
        stage_w(1, WB + 3 * HT);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        MIC_RAW_BARRIER();
        if (wm == 1) MIC_RAW_BARRIER();
        if constexpr (DBG == 3) { tr_c0 = __builtin_amdgcn_s_memtime(); tr_r0 = __builtin_amdgcn_s_memrealtime(); }
        unsigned xs = 0, xprev = 4 * HT, ws = WB;  // slots of tile t and t-1 (= t+2)
        for (int t = 0; t < a.nk; ++t) {
            tile(xs, xs + HT, ws, ws + HT, xprev + HT);
            xprev = xs;
            xs = xs == 4 * HT ? 0 : xs + 2 * HT;
            ws = ws == WB ? WB + 2 * HT : WB;
        }
    } else {
        // prologue: tile 0 and three half-tiles of tile 1
        stage_x(0, OFF_X0);
        stage_w(0, OFF_W0);
        stage_w(1, OFF_W1);
        stage_x(1, OFF_X1);
        advance();
        stage_x(0, KBUF + OFF_X0);
        stage_w(0, KBUF + OFF_W0);
        stage_w(1, KBUF + OFF_W1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        MIC_RAW_BARRIER();
        if (wm == 1) MIC_RAW_BARRIER();
        if constexpr (DBG == 3) { tr_c0 = __builtin_amdgcn_s_memtime(); tr_r0 = __builtin_amdgcn_s_memrealtime(); }
        for (int t = 0; t < a.nk; ++t) {
            const unsigned cur = (t & 1) ? KBUF : 0;
            tile(cur + OFF_X0, cur + OFF_X1, cur + OFF_W0, cur + OFF_W1, KBUF - cur + OFF_X1);
        }
    }
    if (wm == 0) MIC_RAW_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests past the end of K (zeros) are still landing
    __syncthreads();
    if constexpr (DBG == 3) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && a.ln_yt) {
            unsigned long long* o = (unsigned long long*)a.ln_yt + (size_t)blockIdx.x * 8;
            o[0] = tr_c0; o[1] = c1; o[2] = tr_r0; o[3] = r1; o[4] = tr_e0;
        }
    }

    // ---- epilogue through LDS, one wave row (128 pixel rows) at a time: full-row coalesced 16-byte accesses ----
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = i * 16 + (lane & 15), cidx = wn * 64 + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<f32x4_t*>(sC + r * LDC_S + cidx) = acc[j][i];
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 512, r = c >> 5, cc = c & 31;
            const int m = m0 + pass * 128 + r, n = n0 + cc * 8;
            if (m >= a.M || n >= a.N) continue;
            if constexpr (DBG == 8) { if (m > 0) continue; }
            float v[8];
            {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(sC + r * LDC_S + cc * 8);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(sC + r * LDC_S + cc * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            }
            igemm_finish_chunk<false>(a, v, m, n, make_uint4(0, 0, 0, 0));
        }
        if (pass == 0) __syncthreads();
    }
    if constexpr (DBG == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && a.ln_yt) ((unsigned long long*)a.ln_yt + (size_t)blockIdx.x * 8)[5] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

bool mic_igemm8_supported(const IgemmArgs& a) {
    // whole 256-channel tiles (the weight matrix is only padded to 128 rows), K-tiles of 64 elements that never straddle a
    // tap or the boundary between the two K segments, no LayerNorm epilogue
    return a.N % 256 == 0 && a.Kt % 64 == 0 && a.K % 64 == 0 && a.run % 64 == 0 && !a.ln_g;
}

int mic_igemm8_launch(IgemmArgs& a, hipStream_t s, double flops) {
    MIC_ARG_CHECK(mic_igemm8_supported(a), "mic_conv2d: the 256x256 eight-phase variant needs N %% 256 == 0 and K, run %% 64 == 0 (N=%d K=%d run=%d)",
                  a.N, a.Kt, a.run);
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = a.N / 256;
    a.nk = a.Kt / 64;
    const int grid = a.tiles_m * a.tiles_n;
    void* rec = mic_prof_open(s, 1, flops);
    const char* e = getenv("MIC_IGEMM8_DBG");  // measurement builds, see the kernel's comment
    const int dbg = e ? atoi(e) : 0;
    e = getenv("MIC_IGEMM8_LINES");
    const bool lines = e && atoi(e) != 0;
    e = getenv("MIC_IGEMM8_XS3");
    const bool xs3 = e && atoi(e) != 0;
    e = getenv("MIC_IGEMM8_BUFL");
    const bool bufl = e && atoi(e) != 0;
#define MIC_IGEMM8_GO(...) MIC_LAUNCH((igemm8_kernel<__VA_ARGS__>), dim3(grid), dim3(512), 0, s, a)
    if (a.in2) MIC_IGEMM8_GO(true);
    else if (bufl && xs3) MIC_IGEMM8_GO(false, 0, false, true, true);
    else if (bufl && lines) MIC_IGEMM8_GO(false, 0, true, false, true);
    else if (bufl && dbg == 7) MIC_IGEMM8_GO(false, 7, false, false, true);
    else if (bufl && dbg == 8) MIC_IGEMM8_GO(false, 8, false, false, true);
    else if (bufl) MIC_IGEMM8_GO(false, 0, false, false, true);
    else if (xs3 && lines) MIC_IGEMM8_GO(false, 0, true, true);
    else if (xs3 && dbg == 2) MIC_IGEMM8_GO(false, 2, false, true);
    else if (xs3 && dbg == 7) MIC_IGEMM8_GO(false, 7, false, true);
    else if (lines && dbg == 7) MIC_IGEMM8_GO(false, 7, true);
    else if (xs3) MIC_IGEMM8_GO(false, 0, false, true);
    else if (lines && dbg == 2) MIC_IGEMM8_GO(false, 2, true);
    else if (lines) MIC_IGEMM8_GO(false, 0, true);
    else if (dbg == 1) MIC_IGEMM8_GO(false, 1);
    else if (dbg == 2) MIC_IGEMM8_GO(false, 2);
    else if (dbg == 3) MIC_IGEMM8_GO(false, 3);
    else if (dbg == 5) MIC_IGEMM8_GO(false, 5);
    else if (dbg == 6) MIC_IGEMM8_GO(false, 6);
    else if (dbg == 7) MIC_IGEMM8_GO(false, 7);
    else MIC_IGEMM8_GO(false);
#undef MIC_IGEMM8_GO
    return mic_prof_close(rec, s);
}
