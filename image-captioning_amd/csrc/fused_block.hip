// Fused tail of a ResNet bottleneck and head of the next one, bf16, for the identity blocks of the
// early stages where the layer-by-layer trunk is bound by HBM traffic:
//
//   t2  = relu(conv3x3(t1) * W2 + b2)                 [128 pixels x C]     stays in LDS
//   x'  = relu(t2 * W3^T + b3 + x)                    [128 pixels x 4C]    written once (needed as the next residual)
//   t1' = relu(x' * W1n^T + b1n)                      [128 pixels x C2]    written once (input of the next 3x3)
//
// Layer by layer the 4C-wide tensor x' is written by conv3 and read again by the next conv1, and t2 makes a
// round trip of its own; here one workgroup keeps a tile of 128 output pixels on chip from the 3x3
// convolution to the next block's 1x1, so per block the HBM traffic is t1 + x in, x' + t1' out.
//
// One workgroup = 256 threads = 2x2 waves, 128 output pixels.
//   phase 1: implicit-GEMM 3x3 (stride 1, pad 1), K = 9C in 64-byte k-tiles, a ring of three direct-to-LDS stages;
//   phase 2, per chunk of 64 output channels of the 4C: GEMM over K = C from the LDS image of t2 and a staged
//            slice of W3; the epilogue adds bias and the residual slice (staged coalesced into the tile's LDS
//            image beforehand), applies the ReLU, leaves the bf16 tile in LDS and copies it out in full rows;
//   phase 3, same chunk: the tile just produced is 64 of the 4C input channels of the next conv1; its GEMM
//            accumulates over the chunks.
// The weight and residual slices of chunk n+1 are requested into registers while chunk n computes.
// Every LDS operand image is [k-tile][row][64 bytes] with the 16-byte chunks XOR-swizzled as in igemm.hip, so
// a wave's fragment read (ds_read_b128) touches 16 distinct 16-byte slots per lane group.
#include "mic_common.h"

#include <stdlib.h>

namespace {

struct FusedArgs {
    const bf16_t* t1;    // [B][IH][IW][C]
    const bf16_t* w2;    // [C padded to 128][9C], k = (kh*3 + kw)*C + c
    const float* b2;     // [C]
    const bf16_t* w3;    // [4C][C]
    const float* b3;     // [4C]
    const bf16_t* res;   // [M][4C] (no shortcut segment)
    const bf16_t* xs;    // [M][CS]: input of the shortcut convolution (CS > 0), k-columns C .. C+CS-1 of w3
    bf16_t* xout;        // [M][4C]
    const bf16_t* w1n;   // [C2 padded to 128][4C] or nullptr
    const float* b1n;    // [C2]
    bf16_t* t1n;         // [M][C2]
    int M, IH, IW, tiles;
    int dbg;  // timing experiments only (MIC_FUSED_DEBUG): bit 0 skips phase 1, 1 phase 2, 2 its epilogue, 3 the copy-out, 4 phase 3
};

__device__ __forceinline__ int lds_off(int row, int q) { return row * 64 + ((q ^ ((row >> 2) & 3)) << 4); }

typedef unsigned int u32x4_fwd __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma(const u32x4_fwd& wf, const u32x4_fwd& xf, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf), acc, 0, 0, 0);
}

// native vector type for 16-byte moves: aggregate copies of HIP's uint4 struct between address spaces are emitted as
// memcpy through a private stack slot, which puts the register prefetch in scratch memory
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
// buffer descriptor over 2 GiB from p: raw 32-bit byte offsets, bit 31 of an offset = out of range = zeros
__device__ __forceinline__ v4u_t make_srd(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    return v4u_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
}
// One buffer_load_dwordx4 ... lds: every lane fetches 16 bytes at its own byte offset into the buffer, the wave's 1 KiB
// lands at LDS address `lds_base` + 16 * lane.  Inline assembly on purpose: a direct-to-LDS load the compiler can see makes
// it drain the memory counter before the next LDS read; ordering here is the counted s_waitcnt + s_barrier of the loop.
// M0 (the LDS base register) belongs to the compiler, so it is saved and restored around the instruction.
__device__ __forceinline__ void blds16(unsigned voff, const v4u_t& srd, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_base), "s"(srd)
                 : "memory");
}

struct alignas(8) B4 {
    bf16_t v[4];
};

// C: channels of the 3x3 convolution (64 or 128); C2: output channels of the next block's first convolution
// fused behind it (0: none).
// CS: channels of a stride-1 shortcut convolution folded into the expansion as a second K segment (the first block of a
// stage: x' = relu([t2 | x] * [W3 | Wds]^T + b), no residual add); 0: the identity block with its residual tile.
template <int C, int C2, int CS = 0>
__global__ __launch_bounds__(256, (C == 64 && C2 == 64 && CS == 0) ? 3 : 2) void fused_block_kernel(const FusedArgs a) {
    constexpr int BM = 128, NB = 64, NOUT = 4 * C, NCH = NOUT / NB;
    constexpr int KT1 = 9 * C / 32;          // k-tiles of phase 1
    constexpr int KC = C + CS;               // K of phase 2
    constexpr int KT2 = KC / 32;             // k-tiles of phase 2
    constexpr int TN1 = C / 32;              // 16-channel tiles per wave, phase 1 (wave owns C/2 channels)
    constexpr int TN3 = C2 > 0 ? C2 / 32 : 1;
    constexpr int WC1 = C / 64;              // staged weight rows per thread per k-tile, phase 1
    // LDS regions (bytes)
    constexpr int STAGE = (BM + C) * 64;                 // one phase-1 stage: 128 pixel rows then C weight rows
    constexpr int WA = KT2 * NB * 64;                    // W3 slice  [KT2][64 rows][64 B]
    constexpr int WB = C2 > 0 ? 2 * C2 * 64 : 0;         // W1n slice [2][C2 rows][64 B]
    constexpr int A3 = 2 * BM * 64;                      // chunk tile [2][128 rows][64 B]
    constexpr int P23 = WA + WB + A3;
    constexpr int OUT3 = C2 > 0 ? (C2 / 32) * BM * 64 : 0;  // t1' tile [C2/32][128 rows][64 B], over the phase-2/3 region
    constexpr int RING = 3 * STAGE;                      // phase 1 keeps three stages in flight
    constexpr int R0 = (RING > P23 ? RING : P23) > OUT3 ? (RING > P23 ? RING : P23) : OUT3;
    constexpr int A2 = KT2 * BM * 64;                    // t2 tile [KT2][128 rows][64 B]
    __shared__ __attribute__((aligned(16))) unsigned char smem[R0 + A2];
    unsigned char* const sA2 = smem + R0;
    unsigned char* const sWa = smem;
    unsigned char* const sWb = smem + WA;
    unsigned char* const sA3 = smem + WA + WB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // blocks b and b+8 share an XCD: give each XCD a contiguous range of tiles (neighbouring tiles share halo rows)
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int m0 = tile * BM;
    const int HW = a.IH * a.IW;

    // ---------------- phase 1: 3x3 convolution ----------------
    f32x4_t acc1[TN1][4];
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        // Operand stages go global -> LDS directly (buffer_load ... lds, no register staging, no ds_write pass), three
        // stages deep: the requests of k-tile st + 2 are issued while k-tile st is multiplied, one barrier per k-tile, and
        // the memory counter is never drained inside the loop (a workgroup holds 4 waves and a compute unit 2 or 3
        // workgroups: a k-tile's 8 or 16 MFMAs per wave are a fraction of one memory round trip).
        // One wave instruction fills 1 KiB = 16 rows of a stage in lane order; the XOR swizzle of the LDS image is applied
        // to the source chunk a lane fetches: lane l lands in row l >> 2, slot l & 3, which holds chunk slot ^ ((row >> 2) & 3).
        const int q_src = (tid & 3) ^ ((lane >> 4) & 3), row0 = tid >> 2;  // rows row0 and row0 + 64 of the tile (and of W2)
        // Activation rows through a buffer descriptor at 32-bit byte offsets: bit 31 of the offset (a tap in the padding, a row
        // past M; an offset that wrapped below zero has it anyway) makes the hardware return zeros without touching memory.
        const v4u_t srd_x = make_srd(a.t1), srd_w = make_srd(a.w2);
        unsigned bx_off[2], bx_bad[2], bw_off[WC1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + row0 + i * 64;
            const bool ok = m < a.M;
            m = ok ? m : a.M - 1;
            const int b = m / HW, r = m - b * HW, oh = r / a.IW, ow = r - oh * a.IW;
            const int xh = oh - 1, xw = ow - 1;
            bx_off[i] = (unsigned)(((long long)b * HW * C + q_src * 8 + ((long long)xh * a.IW + xw) * C) * 2);
            unsigned bad = ok ? 0u : 0xffffffffu;
#pragma unroll
            for (int th = 0; th < 3; ++th)
#pragma unroll
                for (int tw = 0; tw < 3; ++tw)
                    if (!((unsigned)(xh + th) < (unsigned)a.IH && (unsigned)(xw + tw) < (unsigned)a.IW)) bad |= 1u << (th * 3 + tw);
            bx_bad[i] = bad;
        }
#pragma unroll
        for (int i = 0; i < WC1; ++i) bw_off[i] = (unsigned)(((row0 + i * 64) * (9 * C) + q_src * 8) * 2);
        int kw = 0, c0 = 0;
        unsigned tap = 0, tapb = 0;
        const unsigned rowb = (unsigned)a.IW * C * 2u, pixb = C * 2u;
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
        const unsigned piece = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
        constexpr int NS = RING / STAGE;  // stages in flight; NL loads per thread and stage
        constexpr int NL = 2 + WC1;
        static_assert(NS * STAGE <= R0, "the phase-1 ring fits the region phases 2 and 3 reuse");
        auto request = [&](int slot) __attribute__((always_inline)) {
            const unsigned dst = piece + (unsigned)slot * STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = (bx_off[i] + tapb + (unsigned)c0 * 2u) | (__builtin_amdgcn_ubfe(bx_bad[i], tap, 1) << 31);
                blds16(off, srd_x, dst + i * 64 * 64);
            }
#pragma unroll
            for (int i = 0; i < WC1; ++i) {
                blds16(bw_off[i], srd_w, dst + BM * 64 + i * 64 * 64);
                bw_off[i] += 64;
            }
            c0 += 32;
            if (c0 == C) {
                c0 = 0;
                ++tap;
                tapb += pixb;
                if (++kw == 3) { kw = 0; tapb += rowb - 3u * pixb; }
            }
        };
        int xf_off[4], wf_off[TN1];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf_off[i] = lds_off(wm * 64 + i * 16 + fr, fq);
#pragma unroll
        for (int j = 0; j < TN1; ++j) wf_off[j] = BM * 64 + lds_off(wn * (C / 2) + j * 16 + fr, fq);
        const int nst = (a.dbg & 1) ? 1 : KT1;
        request(0);
        request(1);
        int cur = 0, nxt2 = 2;
        for (int st = 0; st < nst; ++st) {
            // k-tile st has landed for every thread (the next one may still be in flight); every wave is done reading
            // k-tile st - 1, whose slot the requests below overwrite
            if (st + 1 < KT1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (st + 2 < KT1) request(nxt2);
            const unsigned char* base = smem + cur * STAGE;
            u32x4 xf[4], wf[TN1];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4*>(base + xf_off[i]);
#pragma unroll
            for (int j = 0; j < TN1; ++j) wf[j] = *reinterpret_cast<const u32x4*>(base + wf_off[j]);
#pragma unroll
            for (int j = 0; j < TN1; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) mma(wf[j], xf[i], acc1[j][i]);
            cur = cur + 1 == NS ? 0 : cur + 1;
            nxt2 = nxt2 + 1 == NS ? 0 : nxt2 + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (timing builds that cut the loop short leave requests in flight)
    }

    // ---------------- slices of chunk `nc`: W3 rows, W1n columns, residual ----------------
    constexpr int NWA = WA / 16 / 256;                    // 16-byte pieces per thread
    constexpr int NWB = C2 > 0 ? WB / 16 / 256 : 1;
    u32x4 pwa[NWA], pwb[NWB], pres[4];
    auto prefetch = [&](int nc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWA; ++i) {  // the slice is 64 whole rows of W3: one contiguous block
            const int c = tid + 256 * i;
            pwa[i] = *reinterpret_cast<const u32x4*>(a.w3 + (long long)nc * NB * KC + c * 8);
        }
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NWB; ++i) {  // 64 columns (128 bytes) of every row of W1n
                const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
                pwb[i] = *reinterpret_cast<const u32x4*>(a.w1n + (long long)row * NOUT + nc * NB + ch * 8);
            }
        }
        if constexpr (CS == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
                const int m = min(m0 + row, a.M - 1);  // rows past M read the last row; their results are never stored
                pres[i] = *reinterpret_cast<const u32x4*>(a.res + (long long)m * NOUT + nc * NB + ch * 8);
            }
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWA; ++i) {
            const int c = tid + 256 * i, row = c / (KC / 8), ch = c % (KC / 8);
            *reinterpret_cast<u32x4*>(sWa + (ch >> 2) * NB * 64 + lds_off(row, ch & 3)) = pwa[i];
        }
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NWB; ++i) {
                const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
                *reinterpret_cast<u32x4*>(sWb + (ch >> 2) * C2 * 64 + lds_off(row, ch & 3)) = pwb[i];
            }
        }
        if constexpr (CS == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
                *reinterpret_cast<u32x4*>(sA3 + (ch >> 2) * BM * 64 + lds_off(row, ch & 3)) = pres[i];
            }
        }
    };
    prefetch(0);
    if constexpr (CS > 0) {
        // the shortcut's input tile joins t2 as further k-tiles of the expansion's operand image
        constexpr int NX = BM * (CS / 8) / 256;
        u32x4 px[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i, row = c / (CS / 8), ch = c % (CS / 8);
            const int m = min(m0 + row, a.M - 1);
            px[i] = *reinterpret_cast<const u32x4*>(a.xs + (long long)m * CS + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i, row = c / (CS / 8), ch = c % (CS / 8);
            *reinterpret_cast<u32x4*>(sA2 + (C / 32 + (ch >> 2)) * BM * 64 + lds_off(row, ch & 3)) = px[i];
        }
    }

    // ---------------- t2 = relu(acc1 + b2) -> LDS ----------------
#pragma unroll
    for (int j = 0; j < TN1; ++j) {
        const int ch0 = wn * (C / 2) + j * 16 + fq * 4;  // this lane's 4 consecutive channels
        const float4 b = *reinterpret_cast<const float4*>(a.b2 + ch0);
        const int kt = ch0 >> 5, o = ch0 & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 64 + i * 16 + fr;
            B4 p;
            p.v[0] = (bf16_t)fmaxf(acc1[j][i][0] + b.x, 0.f);
            p.v[1] = (bf16_t)fmaxf(acc1[j][i][1] + b.y, 0.f);
            p.v[2] = (bf16_t)fmaxf(acc1[j][i][2] + b.z, 0.f);
            p.v[3] = (bf16_t)fmaxf(acc1[j][i][3] + b.w, 0.f);
            *reinterpret_cast<B4*>(sA2 + kt * BM * 64 + lds_off(row, o >> 3) + (o & 7) * 2) = p;
        }
    }

    // ---------------- phases 2 and 3 over the chunks of 64 output channels ----------------
    f32x4_t acc3[TN3][4];
#pragma unroll
    for (int j = 0; j < TN3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc3[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int x2_off[4], x3_off[4], wa_off[2], wb_off[TN3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x2_off[i] = lds_off(wm * 64 + i * 16 + fr, fq);
        x3_off[i] = lds_off(wm * 64 + i * 16 + fr, fq);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) wa_off[j] = lds_off(wn * 32 + j * 16 + fr, fq);
#pragma unroll
    for (int j = 0; j < TN3; ++j) wb_off[j] = lds_off(wn * (C2 > 0 ? C2 / 2 : 0) + j * 16 + fr, fq);

#pragma unroll
    for (int nc = 0; nc < NCH; ++nc) {
        __syncthreads();  // the previous chunk's readers of Wa / Wb / the tile (and phase 1's of its stages) are done
        stash();
        if (nc + 1 < NCH) prefetch(nc + 1);
        __syncthreads();
        f32x4_t acc2[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (!(a.dbg & 2))
#pragma unroll
        for (int kt = 0; kt < KT2; ++kt) {
            u32x4 xf[4], wf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4*>(sA2 + kt * BM * 64 + x2_off[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const u32x4*>(sWa + kt * NB * 64 + wa_off[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) mma(wf[j], xf[i], acc2[j][i]);
        }
        // epilogue in place: the tile image already holds the residual slice
        if (!(a.dbg & 4))
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = j * 16 + fq * 4;  // channel within this wave's 32 = k within k-tile `wn` of the chunk
            const float4 b = *reinterpret_cast<const float4*>(a.b3 + nc * NB + wn * 32 + o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                B4* cell = reinterpret_cast<B4*>(sA3 + wn * BM * 64 + lds_off(row, o >> 3) + (o & 7) * 2);
                B4 r{};
                if constexpr (CS == 0) r = *cell;
                B4 p;
                p.v[0] = (bf16_t)fmaxf(acc2[j][i][0] + b.x + (float)r.v[0], 0.f);
                p.v[1] = (bf16_t)fmaxf(acc2[j][i][1] + b.y + (float)r.v[1], 0.f);
                p.v[2] = (bf16_t)fmaxf(acc2[j][i][2] + b.z + (float)r.v[2], 0.f);
                p.v[3] = (bf16_t)fmaxf(acc2[j][i][3] + b.w + (float)r.v[3], 0.f);
                *cell = p;
            }
        }
        __syncthreads();
        // copy the finished tile out in full 128-byte row segments
        if (!(a.dbg & 8))
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            const int m = m0 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(sA3 + (ch >> 2) * BM * 64 + lds_off(row, ch & 3));
            if (m < a.M) *reinterpret_cast<u32x4*>(a.xout + (long long)m * NOUT + nc * NB + ch * 8) = v;
        }
        if constexpr (C2 > 0) {
            if (!(a.dbg & 16))
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                u32x4 xf[4], wf[TN3];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4*>(sA3 + kt * BM * 64 + x3_off[i]);
#pragma unroll
                for (int j = 0; j < TN3; ++j) wf[j] = *reinterpret_cast<const u32x4*>(sWb + kt * C2 * 64 + wb_off[j]);
#pragma unroll
                for (int j = 0; j < TN3; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mma(wf[j], xf[i], acc3[j][i]);
            }
        }
    }

    // ---------------- t1' = relu(acc3 + b1n) -> LDS -> full rows ----------------
    if constexpr (C2 > 0) {
        __syncthreads();  // every wave is done with the last chunk's operands: the region is reused for the output tile
#pragma unroll
        for (int j = 0; j < TN3; ++j) {
            const int ch0 = wn * (C2 / 2) + j * 16 + fq * 4;
            const float4 b = *reinterpret_cast<const float4*>(a.b1n + ch0);
            const int kt = ch0 >> 5, o = ch0 & 31;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                B4 p;
                p.v[0] = (bf16_t)fmaxf(acc3[j][i][0] + b.x, 0.f);
                p.v[1] = (bf16_t)fmaxf(acc3[j][i][1] + b.y, 0.f);
                p.v[2] = (bf16_t)fmaxf(acc3[j][i][2] + b.z, 0.f);
                p.v[3] = (bf16_t)fmaxf(acc3[j][i][3] + b.w, 0.f);
                *reinterpret_cast<B4*>(smem + kt * BM * 64 + lds_off(row, o >> 3) + (o & 7) * 2) = p;
            }
        }
        __syncthreads();
        constexpr int CPR = C2 / 8;  // 16-byte pieces per output row
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
            const int m = m0 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (ch >> 2) * BM * 64 + lds_off(row, ch & 3));
            if (m < a.M) *reinterpret_cast<u32x4*>(a.t1n + (long long)m * C2 + ch * 8) = v;
        }
    }
}

template <int C, int C2, int CS = 0> int launch_fused(FusedArgs& a, hipStream_t s) {
    a.tiles = (a.M + 127) / 128;
    static const int dbg = [] { const char* e = getenv("MIC_FUSED_DEBUG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    void* rec = mic_prof_open(s, 1, 2.0 * a.M * ((double)9 * C * C + (double)(C + CS) * 4 * C + (double)4 * C * C2));
    MIC_LAUNCH((fused_block_kernel<C, C2, CS>), dim3(a.tiles), dim3(256), 0, s, a);
    return mic_prof_close(rec, s);
}

}  // namespace

extern "C" int mic_bottleneck_fused_supported(int dtype, int C, int C2) {
    return dtype == MIC_BF16 && ((C == 64 && (C2 == 0 || C2 == 64 || C2 == 128)) || (C == 128 && (C2 == 0 || C2 == 128)));
}

extern "C" int mic_bottleneck_fused_shortcut_supported(int dtype, int C, int C2, int CS) {
    return dtype == MIC_BF16 && C == 64 && C2 == 64 && CS == 64;
}

extern "C" int mic_bottleneck_fused(const mic_fused_block_desc* d, void* stream) {
    MIC_ARG_CHECK(d && d->t1 && d->w2 && d->b2 && d->w3 && d->b3 && d->xout, "mic_bottleneck_fused: null pointer");
    MIC_ARG_CHECK(mic_bottleneck_fused_supported(d->dtype, d->C, d->C2), "mic_bottleneck_fused: unsupported (dtype %d, C=%d, C2=%d)", d->dtype,
                  d->C, d->C2);
    MIC_ARG_CHECK(d->CS == 0 ? d->res != nullptr && d->xs == nullptr
                             : d->res == nullptr && d->xs != nullptr && mic_bottleneck_fused_shortcut_supported(d->dtype, d->C, d->C2, d->CS),
                  "mic_bottleneck_fused: an identity block takes res (CS = 0), a shortcut block xs with a supported CS (got CS=%d)", d->CS);
    MIC_ARG_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && (int64_t)d->B * d->H * d->W < (1ll << 31), "mic_bottleneck_fused: bad geometry %dx%dx%d",
                  d->B, d->H, d->W);
    MIC_ARG_CHECK((int64_t)d->B * d->H * d->W * d->C * 2 < (1ll << 31) - (1 << 20), "mic_bottleneck_fused: t1 must stay below 2 GiB (32-bit byte offsets)");
    MIC_ARG_CHECK(d->C2 == 0 || (d->w1n && d->b1n && d->t1n), "mic_bottleneck_fused: the fused next convolution needs w1n, b1n and t1n");
    MIC_ARG_CHECK(d->w2_rows >= 128 && d->w3_rows >= 4 * d->C && (d->C2 == 0 || d->w1n_rows >= 128),
                  "mic_bottleneck_fused: weight matrices shorter than the kernel reads (w2 %d rows, w3 %d, w1n %d)", d->w2_rows, d->w3_rows,
                  d->w1n_rows);
    const uintptr_t al = (uintptr_t)d->t1 | (uintptr_t)d->w2 | (uintptr_t)d->w3 | (uintptr_t)d->res | (uintptr_t)d->xs | (uintptr_t)d->xout | (uintptr_t)d->b2 |
                         (uintptr_t)d->b3 | (uintptr_t)d->w1n | (uintptr_t)d->b1n | (uintptr_t)d->t1n;
    MIC_ARG_CHECK((al & 15) == 0, "mic_bottleneck_fused: every pointer must be 16-byte aligned");
    MIC_ARG_CHECK(d->res != (const void*)d->xout && d->xs != (const void*)d->xout && d->t1 != (const void*)d->t1n,
                  "mic_bottleneck_fused: outputs must not alias inputs");
    FusedArgs a{};
    a.t1 = (const bf16_t*)d->t1; a.w2 = (const bf16_t*)d->w2; a.b2 = d->b2; a.w3 = (const bf16_t*)d->w3; a.b3 = d->b3;
    a.res = (const bf16_t*)d->res; a.xs = (const bf16_t*)d->xs; a.xout = (bf16_t*)d->xout; a.w1n = (const bf16_t*)d->w1n; a.b1n = d->b1n; a.t1n = (bf16_t*)d->t1n;
    a.M = d->B * d->H * d->W; a.IH = d->H; a.IW = d->W;
    hipStream_t s = (hipStream_t)stream;
    if (d->CS > 0) return launch_fused<64, 64, 64>(a, s);
    if (d->C == 64) {
        if (d->C2 == 0) return launch_fused<64, 0>(a, s);
        if (d->C2 == 64) return launch_fused<64, 64>(a, s);
        return launch_fused<64, 128>(a, s);
    }
    if (d->C2 == 0) return launch_fused<128, 0>(a, s);
    return launch_fused<128, 128>(a, s);
}
