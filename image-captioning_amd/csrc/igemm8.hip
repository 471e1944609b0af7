// Implicit-GEMM convolution / linear layer, bf16, 256 x 256 output tile per workgroup of 8 waves, for the layers with a
// long K (same arithmetic and epilogue as igemm.hip; see there for the operand layout).
//
// Waves are 2 (pixel rows) x 4 (channels): a wave owns 128 pixels x 64 channels = 8 x 4 MFMA tiles, 128 accumulator
// registers.  K advances in tiles of 64 elements.  One K-tile of operands is four half-tiles of 16 KiB in LDS:
//   X0 = pixel rows {0..63} of both wave rows, X1 = rows {64..127} of both, W0 = channels {0..31} of the four wave
//   columns, W1 = channels {32..63}: every wave finds the operands of one quadrant of its output in one X half and one
//   W half.  A half-tile is two planes of 128 rows x 64 bytes in the swizzled image of lds_off().
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
//
// What the instruction stream of the requesting wave costs the MFMA wave it shares a SIMD with decides the speed (both
// issue through one vector port): operands are requested through buffer descriptors at 32-bit byte offsets, one address
// register per lane and three to four vector instructions per pair of requests; a tap outside the image, a row outside
// the problem and a K-tile past the end of K are bit 31 of the offset, for which the hardware returns zeros without
// touching memory.  (With 64-bit pointers and selects the same kernel ran 13-20 % slower.)
//
// Split K: with fewer tiles than compute units, `splitk` workgroups share a tile, each taking a contiguous range of
// K-tiles.  Every one writes its fp32 partial tile to the caller's scratch in fragment order (write-through stores), then
// draws a ticket; the workgroup that draws the last ticket reads all partial tiles back and sums them in slice order
// (so the sum does not depend on which workgroup came last) and runs the epilogue.  The sum is grouped differently from
// the unsplit kernel's, so results agree with it to fp32 rounding, not bit for bit.
#include "igemm_shared.h"

#include <stdlib.h>

namespace {

constexpr int HT = 16384;      // bytes of one half-tile
constexpr int KBUF = 4 * HT;   // one K-tile: X0 | W0 | W1 | X1
constexpr int OFF_X0 = 0, OFF_W0 = HT, OFF_W1 = 2 * HT, OFF_X1 = 3 * HT;
constexpr int SLAB = 256 * 256 * 4;        // one fp32 partial tile
constexpr int COUNTER_BYTES = 4096;        // tickets of up to 1024 tiles at the head of the scratch
constexpr int SC1 = 16;                    // cache-policy bits of a buffer access: write-through store / L1-bypassing load

#define MIC_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// Buffer descriptor of 2 GiB starting at p (raw 32-bit format): loads through it at byte offsets of 2^31 and more return
// zeros instead of touching memory.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u_t make_srd(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    return v4u_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
}
// One buffer_load_dwordx4 ... lds: every lane fetches 16 bytes at its own byte offset into the buffer, the wave's 1 KiB
// lands at LDS address `lds_base` + 16 * lane (see glds16 in igemm_shared.h for why this is inline assembly and how M0 is
// handled).
__device__ __forceinline__ void blds16(unsigned voff, const v4u_t& srd, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_base), "s"(srd)
                 : "memory");
}

// DBG (measurement builds only): 1 = no operand requests inside the loop and 7 = requests only, no LDS reads and no
// MFMAs (both give wrong results); 8 = only output row 0 is stored; 3 = thread 0 of every workgroup stamps the cycle counter
// and the 100 MHz counter around the loop, and the latter at kernel entry and after its last store, into the buffer passed
// as ln_yt (8 x 64-bit words per workgroup)
template <bool SEG2, int DBG = 0> __global__ __launch_bounds__(512) void igemm8_kernel(const IgemmArgs a) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int LDC_S = BN + 4;
    constexpr int EPI = 128 * LDC_S * 4;  // fp32 staging of one wave row's 128 pixel rows
    constexpr int SMEM = EPI > 2 * KBUF ? EPI : 2 * KBUF;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    unsigned long long tr_e0 = 0, tr_c0 = 0, tr_r0 = 0;
    if constexpr (DBG == 3) tr_e0 = __builtin_amdgcn_s_memrealtime();

    // blocks b and b+8 share an XCD (and its L2): every XCD takes a contiguous range of tiles, the slices of a tile together
    int lid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int S = a.splitk, tile_id = lid / S, slice = lid - tile_id * S;
    const int m0 = (tile_id / a.tiles_n) * BM, n0 = (tile_id % a.tiles_n) * BN;
    const int t_begin = (int)((long long)slice * a.nk / S), t_end = (int)((long long)(slice + 1) * a.nk / S);

    // ---- requests: wave w fills rows 16w..16w+15 of every half-tile, both planes (two 1-KiB pieces); the image's swizzle
    // is applied to the 16-byte chunk each lane fetches.  xoff: byte offset of the row's tap (0,0); bit i of xbad: tap i of
    // the row lies outside the image, bit 31: the row lies outside the problem ----
    const int lr = wave * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ (((lr >> 3) & 1) << 1);
    unsigned xoff[2], x2off[2], xbad[2], woff[2];
    {
        const int kh_taps = a.K / a.run / a.kw_taps;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
            const bool ok = m < a.M;
            m = ok ? m : a.M - 1;
            const int b = m / a.OHW, r = m - b * a.OHW, oh = r / a.OW, ow = r - oh * a.OW;
            const int xh = oh * a.stride - a.pad_h, xw = ow * a.stride - a.pad_w;
            xoff[h] = (unsigned)(((long long)b * a.img_stride + (long long)xh * a.row_stride + (long long)xw * a.pix_stride + ch * 8) * 2);
            x2off[h] = SEG2 ? (unsigned)(((long long)b * a.img_stride2 + (long long)(oh * a.stride2) * a.row_stride2 +
                                          (long long)(ow * a.stride2) * a.pix_stride2 + ch * 8) * 2)
                            : 0u;
            unsigned bad = ok ? 0u : 0xffffffffu;
            for (int kh = 0, tp = 0; kh < kh_taps; ++kh)
                for (int kw = 0; kw < a.kw_taps; ++kw, ++tp)
                    if (!((unsigned)(xh + kh) < (unsigned)a.IH && (unsigned)(xw + kw) < (unsigned)a.IW)) bad |= 1u << tp;
            xbad[h] = bad;
            woff[h] = (unsigned)(((long long)(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31)) * a.Kt + (long long)t_begin * BK + ch * 8) * 2);
        }
    }
    const v4u_t srd_x = make_srd(a.in), srd_w = make_srd(a.w), srd_x2 = make_srd(SEG2 ? a.in2 : a.in);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned piece = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);

    // K position of the X requests (the same for every thread): K-tile index, tap, channel offset inside the tap
    int s_t = t_begin, s_c0, s_kh, s_kw, s_tap;
    {
        const int k0 = t_begin * BK;
        s_tap = k0 / a.run;
        s_c0 = k0 - s_tap * a.run;
        s_kh = s_tap / a.kw_taps;
        s_kw = s_tap - s_kh * a.kw_taps;
    }
    unsigned s_offb = (unsigned)(((long long)s_kh * a.row_stride + (long long)s_kw * a.pix_stride + s_c0) * 2);
    auto advance = [&]() {
        ++s_t;
        s_c0 += BK;
        if (s_c0 >= a.run) {
            s_c0 = 0;
            ++s_tap;
            if (++s_kw == a.kw_taps) { s_kw = 0; ++s_kh; }
        }
        s_offb = (unsigned)(((long long)s_kh * a.row_stride + (long long)s_kw * a.pix_stride + s_c0) * 2);
    };
    auto stage_x = [&](int h, unsigned dst) {
        const int k = s_t * BK;
        const bool seg1 = k < a.K;
        if constexpr (DBG == 1) { if (s_t > t_begin + 1) return; }
        const unsigned sb = seg1 ? s_offb : (unsigned)(k - a.K) * 2u;
        const unsigned dead = k < a.Kt ? 0u : 0x80000000u;
        const int bit = seg1 ? s_tap : 31;
        unsigned base = xoff[h];
        if constexpr (SEG2) base = seg1 ? base : x2off[h];
        const unsigned bad = __builtin_amdgcn_ubfe(xbad[h], bit, 1);
        // an offset that wrapped below zero (a tap above or left of the first image) has bit 31 set as well
        const unsigned off = (base + sb) | (bad << 31) | dead;
        if constexpr (SEG2) {
            blds16(off, seg1 ? srd_x : srd_x2, piece + dst);
            blds16(off + 64u, seg1 ? srd_x : srd_x2, piece + dst + 8192u);
        } else {
            blds16(off, srd_x, piece + dst);
            blds16(off + 64u, srd_x, piece + dst + 8192u);
        }
    };
    int w_t[2] = {t_begin, t_begin};
    auto stage_w = [&](int h, unsigned dst) {
        const unsigned dead = w_t[h] * BK < a.Kt ? 0u : 0x80000000u;
        if constexpr (DBG == 1) { if (w_t[h] > t_begin + 1) { ++w_t[h]; return; } }
        blds16(woff[h] | dead, srd_w, piece + dst);
        blds16((woff[h] + 64u) | dead, srd_w, piece + dst + 8192u);
        woff[h] += BK * 2;
        ++w_t[h];
    };

    // ---- fragment reads: lane l holds row (l & 15), 16-byte chunk (l >> 4) of a 16-row x 64-byte sub-tile ----
    const unsigned xrd = (unsigned)lds_off(wm * 64 + (lane & 15), lane >> 4);
    const unsigned wrd = (unsigned)lds_off(wn * 32 + (lane & 15), lane >> 4);
    uint4 xa[4][2] = {}, wb0[2][2] = {}, wb1[2][2] = {};
    auto ldx = [&](unsigned off) {
        if constexpr (DBG == 7) return;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i][kp] = *reinterpret_cast<const uint4*>(smem + off + kp * 8192 + i * 1024 + xrd);
    };
    auto ldw = [&](uint4 (&wb)[2][2], unsigned off) {
        if constexpr (DBG == 7) return;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int j = 0; j < 2; ++j) wb[j][kp] = *reinterpret_cast<const uint4*>(smem + off + kp * 8192 + j * 1024 + wrd);
    };

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

    // one K-tile out of the buffer at `cur`; `oth` is the other buffer
    auto tile = [&](unsigned cur, unsigned oth) {
        // phase 1
        ldw(wb0, cur + OFF_W0);
        __builtin_amdgcn_sched_barrier(0);
        ldx(cur + OFF_X0);
        stage_x(1, oth + OFF_X1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the W0 reads, issued first, have returned
        MIC_RAW_BARRIER();
        quadrant(wb0, 0, 0);
        // phase 2
        ldw(wb1, cur + OFF_W1);
        stage_w(0, cur + OFF_W0);
        __builtin_amdgcn_sched_barrier(0);
        MIC_RAW_BARRIER();
        quadrant(wb1, 0, 1);
        // phase 3
        ldx(cur + OFF_X1);
        advance();
        stage_x(0, cur + OFF_X0);
        __builtin_amdgcn_sched_barrier(0);
        MIC_RAW_BARRIER();
        quadrant(wb1, 1, 1);
        // phase 4
        stage_w(1, cur + OFF_W1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the last three half-tiles: tile t+1 has landed
        MIC_RAW_BARRIER();
        quadrant(wb0, 1, 0);
    };

    // prologue: the first K-tile and three half-tiles of the second
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
    {
        unsigned cur = 0;
        for (int t = t_begin; t < t_end; ++t) {
            tile(cur, KBUF - cur);
            cur = KBUF - cur;
        }
    }
    if (wm == 0) MIC_RAW_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests past the end of the K range are still landing
    __syncthreads();
    if constexpr (DBG == 3) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && a.ln_yt) {
            unsigned long long* o = (unsigned long long*)a.ln_yt + (size_t)blockIdx.x * 8;
            o[0] = tr_c0; o[1] = c1; o[2] = tr_r0; o[3] = r1; o[4] = tr_e0;
        }
    }

    // ---- split K: publish the partial tile, draw a ticket; the last workgroup of the tile sums all of them ----
    if (S > 1) {
        // slab of (tile, slice): register r of thread t at 16 * (512 r + t): every wave-instruction moves 1 KiB contiguously
        const __amdgpu_buffer_rsrc_t slabs = __builtin_amdgcn_make_buffer_rsrc((char*)a.scratch + COUNTER_BYTES, (short)0, (int)0x80000000u, 0x00020000);
        const unsigned slab0 = (unsigned)(tile_id * S) * (unsigned)SLAB + (unsigned)tid * 16u;
        {
            const unsigned mine = slab0 + (unsigned)slice * SLAB;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, acc[j][i]), slabs, mine + (unsigned)(j * 8 + i) * 8192u, 0, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores: complete = visible to every compute unit
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0)
            *flag = __hip_atomic_fetch_add((unsigned*)a.scratch + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        __syncthreads();  // the flag's word belongs to the epilogue staging from here on
        if (ticket != (unsigned)(S - 1)) return;
        if (tid == 0) __hip_atomic_store((unsigned*)a.scratch + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        // every part, this workgroup's own included, read back in slice order with loads that bypass the non-coherent
        // caches: the sum is grouped ((p0 + p1) + p2) ... whichever workgroup came last
        for (int sl = 0; sl < S; ++sl) {
            const unsigned src = slab0 + (unsigned)sl * SLAB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4u_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(slabs, src + (unsigned)(j * 8 + i) * 8192u, 0, SC1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4_t p = __builtin_bit_cast(f32x4_t, v[i]);
                    acc[j][i] = sl == 0 ? p : acc[j][i] + p;
                }
            }
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
    // tap or the boundary between the two K segments, no LayerNorm epilogue; 32-bit byte offsets into tensors of less than
    // 2 GiB and one validity bit per tap
    if (!(a.N % 256 == 0 && a.Kt % 64 == 0 && a.K % 64 == 0 && a.run % 64 == 0 && !a.ln_g)) return false;
    return igemm_fits_buffers(a, 2);
}

extern "C" int64_t mic_conv2d_scratch_bytes(void) { return (int64_t)COUNTER_BYTES + 256ll * SLAB; }

int mic_igemm8_splitk(const IgemmArgs& a) {
    // slices per tile: as many as keep every workgroup resident at once on 256 compute units, at least 24 K-tiles each (with
    // fewer the partial tiles' round trip costs more than the shorter loop saves: measured on 98 tiles, K = 2048 and 4608)
    const int tiles = (a.M + 255) / 256 * (a.N / 256), nk = a.Kt / 64;
    if (!a.scratch || tiles > 128 || tiles > COUNTER_BYTES / 4) return 1;
    int S = 256 / tiles;
    if (S > 4) S = 4;
    while (S > 1 && (nk / S < 24 || (long long)COUNTER_BYTES + (long long)tiles * S * SLAB > a.scratch_bytes)) --S;
    return S;
}

int mic_igemm8_launch(IgemmArgs& a, hipStream_t s, double flops) {
    MIC_ARG_CHECK(mic_igemm8_supported(a),
                  "mic_conv2d: the 256x256 eight-phase variant needs N %% 256 == 0, K and run %% 64 == 0, at most 31 taps and tensors below 2 GiB "
                  "(N=%d K=%d run=%d)", a.N, a.Kt, a.run);
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = a.N / 256;
    a.nk = a.Kt / 64;
    const char* es = getenv("MIC_IGEMM8_SPLITK");  // tuning runs: 0 = never split, n > 1 = n slices wherever the scratch allows
    const int force_split = es ? atoi(es) : -1;
    a.splitk = force_split == 0 ? 1 : mic_igemm8_splitk(a);
    if (force_split > 1 && a.scratch && a.tiles_m * a.tiles_n <= COUNTER_BYTES / 4 && a.nk >= force_split &&
        (long long)COUNTER_BYTES + (long long)a.tiles_m * a.tiles_n * force_split * SLAB <= a.scratch_bytes)
        a.splitk = force_split;
    const int grid = a.tiles_m * a.tiles_n * a.splitk;
    const char* e = getenv("MIC_IGEMM8_DBG");  // measurement builds, see the kernel's comment
    const int dbg = e ? atoi(e) : 0;
    void* rec = mic_prof_open(s, 1, flops);
#define MIC_IGEMM8_GO(...) MIC_LAUNCH((igemm8_kernel<__VA_ARGS__>), dim3(grid), dim3(512), 0, s, a)
    if (a.in2) MIC_IGEMM8_GO(true);
    else if (dbg == 1) MIC_IGEMM8_GO(false, 1);
    else if (dbg == 3) MIC_IGEMM8_GO(false, 3);
    else if (dbg == 7) MIC_IGEMM8_GO(false, 7);
    else if (dbg == 8) MIC_IGEMM8_GO(false, 8);
    else MIC_IGEMM8_GO(false);
#undef MIC_IGEMM8_GO
    return mic_prof_close(rec, s);
}
