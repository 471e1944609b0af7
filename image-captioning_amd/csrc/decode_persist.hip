// One transformer decoder step (every layer and the vocabulary projection for one new position) in ONE launch.
//
// The rows of a step are independent except through the weights: image b's beams only ever read image b's keys, values
// and cache.  So the step is cut into CLUSTERS of 16 images (16 x rows-per-image rows, at most 48: three 16-row matrix
// core tiles).  A cluster is 16 workgroups that never talk to another cluster; inside it every sub-layer is a phase:
//
//   A  packed in-projection q | k | v'      all rows of the cluster x 128 columns per workgroup        (skinny GEMM)
//   B  self-attention, LayerNorm 1, cross-attention, residual      one image per workgroup              (attention)
//   C  feed-forward 1 + ReLU                                        rows x 128 columns                   (skinny GEMM)
//   D  feed-forward 2 + residual (+ row statistics for the next LayerNorm view)                          (skinny GEMM)
//   E  vocabulary projection, after the last layer                                                       (skinny GEMM)
//
// The phases of a layer transpose the work (columns of all rows <-> all columns of an image), so between two phases
// the 16 workgroups exchange their outputs through memory.  That hand-off follows one protocol everywhere:
//   producer: every handed-off byte is stored write-through (buffer_store ... sc1); every wave waits vmcnt(0); workgroup
//             barrier; ONE lane adds 1 to the seam's counter (agent-scope atomic).
//   consumer: ONE lane polls that counter with write-through-aware loads until all 16 have arrived; workgroup barrier;
//             then EVERY load of handed-off bytes is a buffer_load ... sc1 to registers (they bypass the CU's L1, which
//             another CU's stores never refresh).  No such byte is ever read by a plain, scalar or direct-to-LDS load.
// Nothing depends on where a workgroup runs; clusters are formed from workgroups with equal blockIdx % 8 because those
// were observed to share an XCD (a seam measured 2.5 us that way and 14 us across XCDs), which is speed only.
// A workgroup requests everything the NEXT phase needs that no other workgroup produces (its weight fragments, the
// cache rows behind the ancestry table) BEFORE it waits at the seam, so after the seam only the exchanged rows are
// a memory round trip away.  Seam counters only ever grow: a launch reads each counter before it arrives on it and
// takes the multiple of 16 below as its base, so no counter is reset between launches.  Every wait is bounded; a
// timeout raises the workspace's error word and the launch ends.
//
// Arithmetic is that of skinny_gemm_kernel / attn_fused_kernel, through the same helper functions (decode_shared.h):
// the step is held to bit-equality with mic_tr_step by tests/test_gpu_persist.py.
#include "mic_common.h"
#include "decode_shared.h"
#include "search_shared.h"

#include <atomic>

#include <stdlib.h>

namespace {

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int CL = 16;           // workgroups (= images) per cluster
constexpr int NT = 512, NW = 8;  // threads / waves per workgroup
constexpr int SC1 = 16;          // cache-policy bits of a buffer access: write-through store / L1-bypassing load
constexpr int MAX_SEAMS = 64;    // counters per cluster: 4 * layers seams are used
constexpr unsigned SPIN_MAX = 1u << 22;

template <typename T> struct PMma;
template <> struct PMma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf), acc, 0, 0, 0);
    }
};
template <> struct PMma<float> {
    __device__ static __forceinline__ void run(const uint4& wf, const uint4& xf, f32x4_t& acc) {
        const f32x4_t a = __builtin_bit_cast(f32x4_t, wf), b = __builtin_bit_cast(f32x4_t, xf);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
};

// A buffer that workgroups exchange: base and size.  The 128-bit descriptor is built (four scalar moves) where an access
// needs it, so that a phase keeps pointers alive rather than descriptors.
struct HBuf {
    const void* p;
    unsigned bytes;
};
typedef HBuf hbuf_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const hbuf_t& b) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(b.p), (short)0, (int)b.bytes, 0x00020000);
}
// loads of bytes another workgroup of the launch may have written (out-of-range offsets return zero)
__device__ __forceinline__ uint4 hld16(const hbuf_t& b, unsigned off) {
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(mk_rsrc(b), off, 0, SC1);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 hldf4(const hbuf_t& b, unsigned off) { return __builtin_bit_cast(float4, hld16(b, off)); }
// The (mean, M2) pairs of one row, PARTS * 8 contiguous bytes, as 16-byte loads.  (An 8-byte buffer load through
// __builtin_amdgcn_raw_buffer_load_b64 is not used anywhere: hipcc 7.2 emits a single buffer_load_dword for it and the
// second word comes back undefined.)
template <int PARTS> __device__ __forceinline__ void hld_stats(const hbuf_t& b, long long row, float2 (&st)[PARTS]) {
    static_assert(PARTS % 2 == 0, "row statistics are read two parts at a time");
#pragma unroll
    for (int p = 0; p < PARTS; p += 2) {
        const float4 v = hldf4(b, (unsigned)((row * PARTS + p) * 8));
        st[p] = make_float2(v.x, v.y);
        st[p + 1] = make_float2(v.z, v.w);
    }
}
template <bool HAND> __device__ __forceinline__ void hst16(const hbuf_t& b, unsigned off, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v4u_t{v.x, v.y, v.z, v.w}, mk_rsrc(b), off, 0, HAND ? SC1 : 0);
}
template <bool HAND> __device__ __forceinline__ void hst8(const hbuf_t& b, unsigned off, uint2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v2u_t{v.x, v.y}, mk_rsrc(b), off, 0, HAND ? SC1 : 0);
}

struct PLayer {
    const void* qkvf_wf;
    const float* qkvf_b;
    void* kcache;
    void* vfcache;
    const float *self_b, *ln1_g, *ln1_b;
    const void *kf, *vf;
    const float *c, *bo;
    const float *ln2_g, *ln2_b;
    const void* ff1_wf;
    const float* ff1_b;
    const void* ff2_wf;
    const float* ff2_b;
    const float *ln3_g, *ln3_b;
};

struct PersistArgs {
    PLayer layer[MIC_MAX_LAYERS];
    const void* out_wf;
    const float* out_b;
    const float* x;   // [N][H] fp32 residual stream entering the step
    const void* xt;   // [N][H] the same in T
    float* y;         // [N][H] pre-norm sums
    float* qkvf;      // [N][(2 + heads) H]
    void* ffh;        // [N][2H] T
    float* logits;    // [N][V]
    float* stats[2];  // [N][H/128][2]
    const int32_t* anc;
    float* attn;
    unsigned* sync;   // [clusters][MAX_SEAMS] seam counters, then the error word
    int N, B, R, S, t, t_cap, L, V, attn_mode, clusters;
    int stop_seam;  // diagnostic: the launch ends once this seam has been arrived at (-1: never)
    TailArgs tail;    // the beam-search tail of the step as a last phase (has_tail)
    int has_tail;
    unsigned long long* trace;  // diagnostic: [workgroup][128] 100 MHz time stamps, or null
};

// diagnostic time stamp (100 MHz counter) of one workgroup; costs one scalar read and one store when tracing is on
__device__ __forceinline__ void stamp(unsigned long long* trace, int k) {
    if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 128 + k] = __builtin_amdgcn_s_memrealtime();
}

// Workgroup barrier for data exchanged through LDS.  __syncthreads() would also wait for every vector-memory operation
// the wave has in flight (its fence drains the memory counter): the weight fragments, cache rows and folded keys a phase
// has requested ahead of time, and every write-through store, whose completion is a round trip to memory.  Here only the
// LDS counter is waited for.  Nothing in this file hands GLOBAL memory from one wave of a workgroup to another.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- seams ----------------------------------------------------------------------------------------------------
struct Seam {
    unsigned* cnt;    // this cluster's counters
    unsigned* err;    // the launch's error word
    unsigned* base;   // LDS: value each counter had, rounded down to a multiple of 16, when the launch began
    int* abort_flag;  // LDS
};

// end of a phase: every handed-off store of the workgroup has left the CU before the one arrival is counted
__device__ __forceinline__ void seam_arrive(const Seam& s, int idx) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(s.cnt + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Before the first handed-off load of a phase: all 16 workgroups of the cluster have arrived at seam idx.  In two parts,
// so that what a phase requests ahead of its seam sits between them: seam_poll makes the workgroup's FIRST wave (alone)
// wait for the counter, with nothing else in its memory queue (a poll is answered only after every load the wave issued
// before it); the other waves run on and issue their requests; seam_join is the barrier at which they meet.
__device__ __forceinline__ void seam_poll(const Seam& s, int idx, int wave) {
    if (wave == 0 && idx >= 0) {
        if ((threadIdx.x & 63) == 0) {
            const unsigned want = s.base[idx] + CL;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(s.cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > SPIN_MAX || ((spins & 4095u) == 0 && __hip_atomic_load(s.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_store(s.err, 0x100u + (unsigned)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s.abort_flag = 1;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps this wave's loads below the poll
    }
}
__device__ __forceinline__ bool seam_join(const Seam& s) {
    wg_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the handed-off loads below the barrier
    return *s.abort_flag == 0;
}

// ---- GEMM phases ----------------------------------------------------------------------------------------------
template <typename T> struct GemmIO {
    // input rows: T rows (x) or a LayerNorm view of fp32 rows (xraw, xstats, xg, xb), K columns
    bool x_ln;
    hbuf_t x;
    unsigned ld_x_b;
    hbuf_t xraw, xstats;
    const float *xg, *xb;
    const void* wf;     // fragment-packed [N padded to 128][K]
    const float* bias;
    // residual: a LayerNorm view of fp32 rows of N columns, or none
    bool r_ln;
    hbuf_t rraw, rstats;
    const float *rg, *rb;
    hbuf_t out;
    unsigned ldc_b;     // bytes per output row
    float* out_plain;   // the same buffer, for the scalar tail of a row whose width is not a multiple of 4 (fp32 output only)
    bool out_f32;
    int N;              // valid columns
    bool want_stats;
    hbuf_t stats_out;
    int relu, CB, shift;  // column blocks of 128; block cb belongs to workgroup (cb + shift) % 16 of the cluster
};

// One lane's share of a tile's operands that no other workgroup produces: requested before the seam.
constexpr int KH = 16;  // k-tiles of weights a lane holds at a time
struct GemmRegs {
    uint4 w[KH];
    float bv[4], gv[4], bev[4];
    float4 xg[2], xb[2];  // LayerNorm weight and bias of the columns this thread stages (a LayerNorm view as input)
};

// k-tiles [k0, k0 + 16) of this lane's weight fragments (NK: k-tiles of the whole K)
template <typename T, int NK>
__device__ __forceinline__ void gemm_load_w(uint4 (&w)[KH], const GemmIO<T>& io, int cb, int wave, int lane, int k0) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int nb = cb * 128 + wave * 16;
    const T* wp = (const T*)io.wf + ((long long)(nb / 16) * NK * 64 + lane) * VEC;
#pragma unroll
    for (int kt = 0; kt < KH; ++kt) w[kt] = *reinterpret_cast<const uint4*>(wp + (long long)(k0 + kt) * 64 * VEC);
}

template <typename T, int NK>
__device__ __forceinline__ void gemm_prefetch(GemmRegs& g, const GemmIO<T>& io, int cb, int wave, int lane) {
    constexpr int VEC_ = 16 / (int)sizeof(T), CPR_ = 4 * NK;
    static_assert(NT % CPR_ == 0, "a thread stages the same 16-byte column chunk of every row it handles");
    if (io.x_ln) {
        const int ch = (wave * 64 + lane) % CPR_;
#pragma unroll
        for (int e4 = 0; e4 < VEC_ / 4; ++e4) {
            g.xg[e4] = *reinterpret_cast<const float4*>(io.xg + ch * VEC_ + e4 * 4);
            g.xb[e4] = *reinterpret_cast<const float4*>(io.xb + ch * VEC_ + e4 * 4);
        }
    }
    const int q = lane >> 4;
    const int nb = cb * 128 + wave * 16;
    gemm_load_w<T, NK>(g.w, io, cb, wave, lane, 0);
    const int n = nb + q * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = n + e < io.N;
        g.bv[e] = ok && io.bias ? io.bias[n + e] : 0.f;
        g.gv[e] = ok && io.r_ln ? io.rg[n + e] : 0.f;
        g.bev[e] = ok && io.r_ln ? io.rb[n + e] : 0.f;
    }
}

// Row statistics of the cluster's rows (LayerNorm views) into LDS, then the rows themselves into LDS in T, K columns.
// Every thread of the workgroup calls; ends with a barrier.
template <typename T, int MT, int NK, int PARTS>
__device__ __forceinline__ void gemm_stage_x(const GemmRegs& g, const GemmIO<T>& io, int m0, int M, unsigned char* xs, float (*rs_x)[2],
                                             float (*rs_r)[2], int wave, int lane, unsigned long long* trace, int tk) {
    constexpr int VEC = 16 / (int)sizeof(T), ROWS = MT * 16, CPR = 4 * NK, K = CPR * VEC, XROW = K * (int)sizeof(T) + 16;
    constexpr int XCH = ROWS * CPR / NT;  // 16-byte chunks per thread
    constexpr int RSTEP = NT / CPR;       // rows between two chunks of a thread
    static_assert(ROWS * CPR % NT == 0 && NT % CPR == 0, "whole chunks per thread, one column chunk per thread");
    const int tid = wave * 64 + lane;
    const int ch = tid % CPR, row0 = tid / CPR;
    // the row statistics (first 16 * MT threads) and the rows are requested together: one memory round trip for both
    float2 px[PARTS], pr[PARTS];
    if (tid < ROWS) {
        const int mr = m0 + tid < M ? m0 + tid : M - 1;
        if (io.x_ln) hld_stats<PARTS>(io.xstats, mr, px);
        if (io.r_ln) hld_stats<PARTS>(io.rstats, mr, pr);
    }
    auto merge_stats = [&]() {
        if (tid < ROWS) {
            if (io.x_ln) {
                float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
                for (int p = 0; p < PARTS; ++p) chan_step(n, mu, m2, 128.f, px[p].x, px[p].y);
                rs_x[tid][0] = mu;
                rs_x[tid][1] = chan_rstd(m2, n);
            }
            if (io.r_ln) {
                float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
                for (int p = 0; p < PARTS; ++p) chan_step(n, mu, m2, 128.f, pr[p].x, pr[p].y);
                rs_r[tid][0] = mu;
                rs_r[tid][1] = chan_rstd(m2, n);
            }
            stamp(trace, tk);
        }
    };
    if (io.x_ln) {
        float4 raw[XCH][VEC / 4];
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int row = row0 + i * RSTEP;
            const int mr = m0 + row < M ? m0 + row : M - 1;
#pragma unroll
            for (int e4 = 0; e4 < VEC / 4; ++e4) raw[i][e4] = hldf4(io.xraw, (unsigned)((mr * K + ch * VEC + e4 * 4) * 4));
        }
        merge_stats();
        wg_barrier();
        if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(trace, tk + 1); }
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int row = row0 + i * RSTEP;
            const float mean = rs_x[row][0], rstd = rs_x[row][1];
            Vec16<T> o;
#pragma unroll
            for (int e4 = 0; e4 < VEC / 4; ++e4) {
                const float4 v = raw[i][e4];
                const float4 gg = g.xg[e4], b = g.xb[e4];
                o.v[e4 * 4 + 0] = from_f32<T>(ln_apply(v.x, mean, rstd, gg.x, b.x));
                o.v[e4 * 4 + 1] = from_f32<T>(ln_apply(v.y, mean, rstd, gg.y, b.y));
                o.v[e4 * 4 + 2] = from_f32<T>(ln_apply(v.z, mean, rstd, gg.z, b.z));
                o.v[e4 * 4 + 3] = from_f32<T>(ln_apply(v.w, mean, rstd, gg.w, b.w));
            }
            st16(reinterpret_cast<T*>(xs + row * XROW + ch * 16), o);
        }
    } else {
        uint4 raw[XCH];
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int row = row0 + i * RSTEP;
            const int mr = m0 + row < M ? m0 + row : M - 1;
            raw[i] = hld16(io.x, (unsigned)mr * io.ld_x_b + (unsigned)ch * 16u);
        }
        merge_stats();
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int row = row0 + i * RSTEP;
            *reinterpret_cast<uint4*>(xs + row * XROW + ch * 16) = raw[i];
        }
    }
    wg_barrier();
    stamp(trace, tk + 2);
}

// The MFMAs and the epilogue of one tile (MT * 16 rows x 128 columns) from staged rows and prefetched weights.
template <typename T, int MT, int NK, bool HAND>
__device__ __forceinline__ void gemm_tile(const GemmRegs& g, const uint4 (&w2)[KH], const GemmIO<T>& io, int cb, int m0, int M, const unsigned char* xs,
                                          const float (*rs_r)[2], float (*red)[NW][48], int wave, int lane, unsigned long long* trace, int tk) {
    constexpr int VEC = 16 / (int)sizeof(T), CPR = 4 * NK, K = CPR * VEC, XROW = K * (int)sizeof(T) + 16, ROWS = MT * 16;
    const int tid = wave * 64 + lane, r = lane & 15, q = lane >> 4;
    const int nb = cb * 128 + wave * 16, n = nb + q * 4;
    // residual values of this lane's outputs (rows of N columns, N a multiple of 128 for a LayerNorm view)
    float4 rv[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (io.r_ln) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + i * 16 + r, mc = m < M ? m : M - 1;
            rv[i] = hldf4(io.rraw, (unsigned)((mc * io.N + n) * 4));
        }
    }
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* xp = xs + r * XROW + q * 16;
    static_assert(NK == KH || NK == 2 * KH, "K is 16 or 32 k-tiles");
    // The fragment reads of four k-tiles are issued together, ahead of their MFMAs (left to itself the compiler, short
    // of registers, waits for every read right before the one MFMA that takes it).
    constexpr int KG = 4;
    auto mma_half = [&](const uint4 (&w)[KH], int kbase) {
#pragma unroll
        for (int k0 = 0; k0 < KH; k0 += KG) {
            uint4 xf[KG][MT];
#pragma unroll
            for (int u = 0; u < KG; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[u][i] = *reinterpret_cast<const uint4*>(xp + i * 16 * XROW + (kbase + k0 + u) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KG; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) PMma<T>::run(w[k0 + u], xf[u][i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    mma_half(g.w, 0);
    if constexpr (NK > KH) mma_half(w2, KH);
    if (trace) { asm volatile("" :: "v"(acc[0][0])); stamp(trace, tk + 3); }
    float v[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float rr4[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[i][e] + g.bv[e];
            if (n + e < io.N) {
                if (io.r_ln) x += ln_apply(rr4[e], rs_r[i * 16 + r][0], rs_r[i * 16 + r][1], g.gv[e], g.bev[e]);
            } else {
                x = 0.f;
            }
            v[i][e] = io.relu ? fmaxf(x, 0.f) : x;
        }
    }
    if (trace) { asm volatile("" :: "v"(v[0][0])); stamp(trace, tk + 4); }
    if (io.want_stats) {
        // per row: mean and centred sum of squares over this workgroup's 128 columns (valid ones)
        const float cnt = (float)min(128, io.N - cb * 128);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            s = sum_xor16(s);
            s = sum_xor32(s);
            if (q == 0) red[0][wave][i * 16 + r] = s;
        }
        wg_barrier();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = i * 16 + r;
            float rsum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) rsum += red[0][w8][row];
            const float mean = rsum / cnt;
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = n + e < io.N ? v[i][e] - mean : 0.f;
                sq = fmaf(d, d, sq);
            }
            sq = sum_xor16(sq);
            sq = sum_xor32(sq);
            if (q == 0) red[1][wave][row] = sq;
        }
        wg_barrier();
        float st_mean = 0.f, st_m2 = 0.f;
        if (tid < ROWS && m0 + tid < M) {
            float rsum = 0.f, m2 = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) { rsum += red[0][w8][tid]; m2 += red[1][w8][tid]; }
            st_mean = rsum / cnt;
            st_m2 = m2;
        }
        wg_barrier();  // the scratch is free for the workgroup's next tile
        if (tid < ROWS && m0 + tid < M)
            hst8<HAND>(io.stats_out, (unsigned)(((m0 + tid) * io.CB + cb) * 8), make_uint2(__builtin_bit_cast(unsigned, st_mean), __builtin_bit_cast(unsigned, st_m2)));
        stamp(trace, tk + 5);
    }
    if (n < io.N) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + i * 16 + r;
            if (m >= M) continue;
            if (io.out_f32) {
                if (n + 3 < io.N && (io.ldc_b & 15u) == 0) {
                    hst16<HAND>(io.out, (unsigned)m * io.ldc_b + (unsigned)n * 4u,
                                make_uint4(__builtin_bit_cast(unsigned, v[i][0]), __builtin_bit_cast(unsigned, v[i][1]),
                                           __builtin_bit_cast(unsigned, v[i][2]), __builtin_bit_cast(unsigned, v[i][3])));
                } else {  // ragged row end or unaligned rows: never a handed-off buffer (checked on the host)
                    float* o = io.out_plain + (size_t)m * (io.ldc_b >> 2) + n;
                    for (int e = 0; e < 4; ++e)
                        if (n + e < io.N) o[e] = v[i][e];
                }
            } else {
                struct alignas(8) B4 { bf16_t v[4]; } p;
#pragma unroll
                for (int e = 0; e < 4; ++e) p.v[e] = (bf16_t)v[i][e];
                hst8<HAND>(io.out, (unsigned)m * io.ldc_b + (unsigned)n * 2u, __builtin_bit_cast(uint2, p));
            }
        }
    }
}

// One GEMM phase of the cluster.  `wait_idx` < 0: the phase reads nothing another workgroup of the launch wrote.
// Returns false when a wait timed out.
template <typename T, int MT, int NK, int PARTS, bool HAND>
__device__ __forceinline__ bool gemm_phase(const GemmIO<T>& io, const Seam& seam, int wait_idx, int j, int m0, int M, unsigned char* xs,
                                           float (*rs_x)[2], float (*rs_r)[2], float (*red)[NW][48], int wave, int lane,
                                           unsigned long long* trace) {
    bool first = true;
    for (int cb = (j - io.shift + 2 * CL) % CL; cb < io.CB; cb += CL) {
        GemmRegs g;
        uint4 w2[KH];  // second half of a K of 32 k-tiles: requested once the seam is passed, in flight while the rows are staged
        if (first) seam_poll(seam, wait_idx, wave);
        gemm_prefetch<T, NK>(g, io, cb, wave, lane);
        if constexpr (NK > KH) { if (first) gemm_load_w<T, NK>(w2, io, cb, wave, lane, KH); }
        if (first) {
            if (!seam_join(seam)) return false;
            stamp(trace, 3 * (wait_idx + 1) + 1);
            gemm_stage_x<T, MT, NK, PARTS>(g, io, m0, M, xs, rs_x, rs_r, wave, lane, trace, 40 + 6 * (wait_idx + 1));
            first = false;
        } else if constexpr (NK > KH) {
            gemm_load_w<T, NK>(w2, io, cb, wave, lane, KH);
        }
        gemm_tile<T, MT, NK, HAND>(g, w2, io, cb, m0, M, xs, rs_r, red, wave, lane, trace, 40 + 6 * (wait_idx + 1));
    }
    return true;
}

// A 16-byte load of data that is read once per step and never again before most of the caches have turned over (the
// self-attention cache rows, the folded cross keys / values): flagged non-temporal when MIC_PERSIST_NT is set at build time.
template <typename T> __device__ __forceinline__ Vec16<T> ld16s(const T* p) {
#ifdef MIC_PERSIST_NT
    typedef unsigned v4u_nt __attribute__((ext_vector_type(4)));
    const v4u_nt r = __builtin_nontemporal_load(reinterpret_cast<const v4u_nt*>(p));
    return __builtin_bit_cast(Vec16<T>, r);
#else
    return ld16(p);
#endif
}

// ---- attention phase ------------------------------------------------------------------------------------------
// PPW: cached positions per wave and row (4: up to 32 positions over the 8 waves, 6: up to 48).  With 6 the folded values of
// a wave's last three positions are requested only once its keys have been consumed (their registers).
template <typename T, int RG, int HEADS, int PPW> struct AttnPre {
    static constexpr int NJ = RG * PPW, PPE = PPW == 4 ? 4 : 3, PPL = PPW - PPE;
    int srow[NJ];  // wave-uniform: the row whose cache holds position p of this wave's pair
    Vec16<T> kreg[NJ], vreg[RG * PPE][HEADS];
};

struct AttnIO {
    hbuf_t qkvf;        // handed off
    bool x_plain;       // residual stream: x rows (previous launch) or the LayerNorm view below (handed off)
    hbuf_t xr, xstats;
    const float *xg, *xb;
    hbuf_t y, stats_out;  // outputs, handed off
    float* attn_out;
    int ld_attn, accumulate;
    float attn_scale;
};

// cache rows of this wave's (row, position) pairs: ancestry, then keys and folded values.  Nothing here was written
// by another workgroup of this launch: the cache cells of earlier positions and the ancestry table come from earlier launches.
template <typename T, int RG, int HEADS, int PPW>
__device__ __forceinline__ void attn_ancestry(AttnPre<T, RG, HEADS, PPW>& pre, const int32_t* anc, long long n0, int rn, int t, int t_cap, int wave) {
    constexpr int NJ = RG * PPW;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int rr = jj / PPW, p = wave + NW * (jj % PPW);
        const int own = (int)n0 + (rr < rn ? rr : 0);
        pre.srow[jj] = own;
        if (anc) {
            const int got = __builtin_amdgcn_readfirstlane(anc[(long long)own * t_cap + (p < t_cap ? p : 0)]);
            pre.srow[jj] = p < t ? got : own;  // entries at and beyond t are not ancestry yet
        }
    }
}
template <typename T, int RG, int HEADS, int PPW>
__device__ __forceinline__ void attn_prefetch(AttnPre<T, RG, HEADS, PPW>& pre, const PLayer& ly, int t, int t_cap, int wave, int lane) {
    constexpr int VEC = Vec16<T>::N, H = 64 * VEC, NJ = RG * PPW, PPE = AttnPre<T, RG, HEADS, PPW>::PPE;
    const int lc = lane * VEC;
    const T* kc = (const T*)ly.kcache;
    const T* vc = (const T*)ly.vfcache;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int q = jj % PPW, p = wave + NW * q;
        const long long cell = (long long)pre.srow[jj] * t_cap + (p < t ? p : 0);
        pre.kreg[jj] = ld16s(kc + cell * H + lc);
        if (q < PPE) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h) pre.vreg[(jj / PPW) * PPE + q][h] = ld16s(vc + (cell * HEADS + h) * H + lc);
        }
    }
}

// Both attention sub-layers of image b (rows n0 .. n0 + rn): attn_fused_kernel with its exchanged operands read
// through write-through-aware loads and its outputs stored write-through.
template <typename T, int RG, int HEADS, int PPW>
__device__ __forceinline__ void attn_body(AttnPre<T, RG, HEADS, PPW>& pre, const AttnIO& io, const PLayer& ly, int b, long long n0, int rn, int S,
                                          int t, int t_cap, float* smem, int wave, int lane) {
    constexpr int VEC = Vec16<T>::N, H = 64 * VEC, PARTS = H / 128, SP = 64;
    constexpr int NJ = RG * PPW, NJP = pow2_at_least(NJ), PPE = AttnPre<T, RG, HEADS, PPW>::PPE, PPL = PPW - PPE;
    constexpr int NI = 64 / NW, NX = NI * HEADS;
    constexpr int F4 = H / 4, NRV = (RG * F4 + NT - 1) / NT;
    constexpr int QW = (2 + HEADS) * H, LPH = 64 / HEADS;
    float* xin = smem;                   // [RG][H] residual stream entering the layer
    float* xs = xin + RG * H;            // [RG][H] LayerNorm1(y1)
    float* qs = xs + RG * H;             // [RG][H] scaled queries
    float* sc = qs + RG * H;             // [RG][HEADS][SP]
    float* part = sc + RG * HEADS * SP;  // [NW][RG][H]
    float* vecs = part + NW * RG * H;    // [3][H]
    const int tid = wave * 64 + lane, P = t + 1;
    const int lc = lane * VEC, hl = lane / LPH;
    const float scale = sqrtf(1.0f / (float)(H / HEADS));
    const T* kfb = (const T*)ly.kf + (long long)b * S * HEADS * H;
    const T* vfb = (const T*)ly.vf + (long long)b * S * HEADS * H;

    if (tid < 3 * F4) {
        const int which = tid / F4, c4 = (tid - which * F4) * 4;
        const float* src = which == 0 ? ly.self_b : which == 1 ? ly.ln1_g : ly.ln1_b;
        *reinterpret_cast<float4*>(vecs + which * H + c4) = *reinterpret_cast<const float4*>(src + c4);
    }
    // residual rows and queries (one float4 per thread and round), with the statistics of the row when it is a view
    float4 rv[NRV], gv[NRV], bv[NRV], qv[NRV];
    float2 stp[NRV][PARTS];
#pragma unroll
    for (int u = 0; u < NRV; ++u) {
        const int i = tid + NT * u, rr = i < rn * F4 ? i / F4 : 0, c4 = (i - (i / F4) * F4) * 4;
        rv[u] = hldf4(io.xr, (unsigned)(((n0 + rr) * H + c4) * 4));
        qv[u] = hldf4(io.qkvf, (unsigned)(((n0 + rr) * QW + c4) * 4));
        if (!io.x_plain) {
            gv[u] = *reinterpret_cast<const float4*>(io.xg + c4);
            bv[u] = *reinterpret_cast<const float4*>(io.xb + c4);
            hld_stats<PARTS>(io.xstats, n0 + rr, stp[u]);
        }
    }
    // the new position of row w, by wave w: the fresh projection rounded through T, exactly what the cache will hold
    Vec16<T> knew, vnew[HEADS];
    if (wave < rn) {
        const unsigned fo = (unsigned)(((n0 + wave) * QW + H + lc) * 4);
        const long long cell = (n0 + wave) * t_cap + t;
        float f[(1 + HEADS) * VEC];
#pragma unroll
        for (int hh = 0; hh < 1 + HEADS; ++hh)
#pragma unroll
            for (int e4 = 0; e4 < VEC; e4 += 4) {
                const float4 w4 = hldf4(io.qkvf, fo + (unsigned)((hh * H + e4) * 4));
                f[hh * VEC + e4] = w4.x; f[hh * VEC + e4 + 1] = w4.y; f[hh * VEC + e4 + 2] = w4.z; f[hh * VEC + e4 + 3] = w4.w;
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) knew.v[e] = from_f32<T>(f[e]);
        st16((T*)ly.kcache + cell * H + lc, knew);
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) vnew[h].v[e] = from_f32<T>(f[(1 + h) * VEC + e]);
            st16((T*)ly.vfcache + (cell * HEADS + h) * H + lc, vnew[h]);
        }
    }
    // ---- residual stream and scaled queries of the rows -> LDS ----
#pragma unroll
    for (int u = 0; u < NRV; ++u) {
        const int i = tid + NT * u, rr = i / F4, c4 = (i - rr * F4) * 4;
        if (i < rn * F4) {
            float4 v = rv[u];
            if (!io.x_plain) {
                float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
                for (int p = 0; p < PARTS; ++p) chan_step(n, mu, m2, 128.f, stp[u][p].x, stp[u][p].y);
                const float rstd = chan_rstd(m2, n);
                v = make_float4(ln_apply(v.x, mu, rstd, gv[u].x, bv[u].x), ln_apply(v.y, mu, rstd, gv[u].y, bv[u].y),
                                ln_apply(v.z, mu, rstd, gv[u].z, bv[u].z), ln_apply(v.w, mu, rstd, gv[u].w, bv[u].w));
            }
            *reinterpret_cast<float4*>(xin + rr * H + c4) = v;
            *reinterpret_cast<float4*>(qs + rr * H + c4) = make_float4(qv[u].x * scale, qv[u].y * scale, qv[u].z * scale, qv[u].w * scale);
        }
    }
    wg_barrier();
    // ---- self-attention scores ----
    {
        float pv[NJP];
#pragma unroll
        for (int jj = 0; jj < NJP; ++jj) {
            pv[jj] = 0.f;
            if (jj < NJ) {
                const float* qp = qs + (jj / PPW) * H + lc;
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4) {
                    const float4 qq = *reinterpret_cast<const float4*>(qp + e4);
                    pv[jj] = fmaf(qq.x, to_f32<T>(pre.kreg[jj].v[e4]), pv[jj]);
                    pv[jj] = fmaf(qq.y, to_f32<T>(pre.kreg[jj].v[e4 + 1]), pv[jj]);
                    pv[jj] = fmaf(qq.z, to_f32<T>(pre.kreg[jj].v[e4 + 2]), pv[jj]);
                    pv[jj] = fmaf(qq.w, to_f32<T>(pre.kreg[jj].v[e4 + 3]), pv[jj]);
                }
            }
        }
        bool writer;
        const int jj = reduce_scatter<NJP, LPH>(pv, lane, writer);
        const int rr = jj / PPW, p = wave + NW * (jj % PPW);
        if (writer && jj < NJ && rr < rn && p < t) sc[(rr * HEADS + hl) * SP + p] = pv[0];
    }
    if (wave < rn) {
        const float* qp = qs + wave * H + lc;
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v = fmaf(qp[e], to_f32<T>(knew.v[e]), v);
        v = group_sum_down<LPH>(v);
        if ((lane & (LPH - 1)) == 0) sc[(wave * HEADS + hl) * SP + t] = v;
    }
    // the folded values of the late positions, then the folded cross keys (and their constants): requested once the
    // self-attention keys have been consumed
    Vec16<T> vlate[PPL > 0 ? RG * PPL : 1][HEADS];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int q = jj % PPW, p = wave + NW * q;
        if (q >= PPE) {
            const long long cell = (long long)pre.srow[jj] * t_cap + (p < t ? p : 0);
#pragma unroll
            for (int h = 0; h < HEADS; ++h)
                vlate[(jj / PPW) * PPL + q - PPE][h] = ld16s((const T*)ly.vfcache + (cell * HEADS + h) * H + lc);
        }
    }
    Vec16<T> kx[NI][HEADS];
    float cx[NI][HEADS];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + NW * i, pc = p < S ? p : S - 1;  // tokens past S: a valid row, masked when used
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            kx[i][h] = ld16s(kfb + (pc * HEADS + h) * H + lc);
            cx[i][h] = ly.c[((long long)b * S + pc) * HEADS + h];
        }
    }
    wg_barrier();
    for (int rh = wave; rh < rn * HEADS; rh += NW) {
        float* row = sc + rh * SP;
        const float pr = wave_softmax(lane < P ? row[lane] : 0.f, lane < P);
        if (lane < P) row[lane] = pr;
    }
    wg_barrier();
    // ---- self-attention output, one row at a time: this wave's share -> LDS ----
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int p = wave + NW * q;
            const bool on = rr < rn && p < t;
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                // a masked pair was loaded from a cell nothing has written yet: its bytes are selected away, not multiplied by zero
                const Vec16<T>& held = q < PPE ? pre.vreg[rr * PPE + q][h] : vlate[PPL > 0 ? rr * PPL + q - PPE : 0][h];
                const uint4 bits = on ? __builtin_bit_cast(uint4, held) : make_uint4(0u, 0u, 0u, 0u);
                const Vec16<T> val = __builtin_bit_cast(Vec16<T>, bits);
                const float wgt = on ? sc[(rr * HEADS + h) * SP + p] : 0.f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(val.v[e]), acc[e]);
            }
        }
        if (wave == rr && rr < rn) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                const float wgt = sc[(rr * HEADS + h) * SP + t];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(vnew[h].v[e]), acc[e]);
            }
        }
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4)
            *reinterpret_cast<float4*>(part + (wave * RG + rr) * H + lc + e4) = make_float4(acc[e4], acc[e4 + 1], acc[e4 + 2], acc[e4 + 3]);
    }
    // folded cross values, requested once the self-attention values have been consumed
    Vec16<T> vx[NI][HEADS];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + NW * i, pc = p < S ? p : S - 1;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) vx[i][h] = ld16s(vfb + (pc * HEADS + h) * H + lc);
    }
    wg_barrier();
    // ---- y1 = x + attention + b', u = LayerNorm1(y1): one wave per row ----
    if (wave < rn) {
        const int rr = wave;
        float v[VEC];
        float s = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4) {
            float4 o = *reinterpret_cast<const float4*>(part + rr * H + lc + e4);
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float4 pw = *reinterpret_cast<const float4*>(part + (w * RG + rr) * H + lc + e4);
                o.x += pw.x; o.y += pw.y; o.z += pw.z; o.w += pw.w;
            }
            const float4 xi = *reinterpret_cast<const float4*>(xin + rr * H + lc + e4);
            const float4 sb = *reinterpret_cast<const float4*>(vecs + lc + e4);
            v[e4] = o.x + sb.x + xi.x; v[e4 + 1] = o.y + sb.y + xi.y;
            v[e4 + 2] = o.z + sb.z + xi.z; v[e4 + 3] = o.w + sb.w + xi.w;
            s += (v[e4] + v[e4 + 1]) + (v[e4 + 2] + v[e4 + 3]);
        }
        const float mean = wave_sum(s) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] -= mean;
            sq = fmaf(v[e], v[e], sq);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + 1e-5f);
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4) {
            const float4 g1 = *reinterpret_cast<const float4*>(vecs + H + lc + e4), b1 = *reinterpret_cast<const float4*>(vecs + 2 * H + lc + e4);
            *reinterpret_cast<float4*>(xs + rr * H + lc + e4) =
                make_float4(ln_scale(v[e4], rstd, g1.x, b1.x), ln_scale(v[e4 + 1], rstd, g1.y, b1.y), ln_scale(v[e4 + 2], rstd, g1.z, b1.z),
                            ln_scale(v[e4 + 3], rstd, g1.w, b1.w));
        }
    }
    wg_barrier();
    // ---- cross-attention scores u . K'_h[s] + c_h[s], one row at a time ----
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        float qx[VEC];
        const float* qp = xs + (rr < rn ? rr : 0) * H + lc;
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + e4);
            qx[e4] = v.x; qx[e4 + 1] = v.y; qx[e4 + 2] = v.z; qx[e4 + 3] = v.w;
        }
        float pv[NX];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                float v = lane == 0 ? cx[i][h] : 0.f;  // the constant rides on one lane's share of the sum
#pragma unroll
                for (int e = 0; e < VEC; ++e) v = fmaf(qx[e], to_f32<T>(kx[i][h].v[e]), v);
                pv[i * HEADS + h] = v;
            }
        bool writer;
        const int x = reduce_scatter<NX, 64>(pv, lane, writer);
        const int p = wave + NW * (x / HEADS);
        if (writer && rr < rn && p < S) sc[(rr * HEADS + x % HEADS) * SP + p] = pv[0];
    }
    wg_barrier();
    for (int rh = wave; rh < rn * HEADS; rh += NW) {
        float* row = sc + rh * SP;
        const float pr = wave_softmax(lane < S ? row[lane] : 0.f, lane < S);
        if (lane < S) row[lane] = pr;
    }
    wg_barrier();
    if (io.attn_out) {
        const float hs = io.attn_scale / (float)HEADS;
        for (int i = tid; i < rn * S; i += NT) {
            const int rr = i / S, p = i - rr * S;
            float sum = 0.f;
#pragma unroll
            for (int hh = 0; hh < HEADS; ++hh) sum += sc[(rr * HEADS + hh) * SP + p];
            float* dst = io.attn_out + (n0 + rr) * io.ld_attn + p;
            *dst = __builtin_fmaf(sum, hs, io.accumulate ? *dst : 0.f);
        }
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + NW * i;
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                const float wgt = rr < rn && p < S ? sc[(rr * HEADS + h) * SP + p] : 0.f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(vx[i][h].v[e]), acc[e]);
            }
        }
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4)
            *reinterpret_cast<float4*>(part + (wave * RG + rr) * H + lc + e4) = make_float4(acc[e4], acc[e4 + 1], acc[e4 + 2], acc[e4 + 3]);
    }
    wg_barrier();
    // ---- y2 = u + attention + b_o and its statistics per 128 columns: one wave per (row, part), 2 columns per lane ----
    for (int pair = wave; pair < rn * PARTS; pair += NW) {
        const int rr = pair / PARTS, pp = pair - rr * PARTS;
        const int col = pp * 128 + lane * 2;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float2 pw = *reinterpret_cast<const float2*>(part + (w * RG + rr) * H + col);
            v0 += pw.x; v1 += pw.y;
        }
        const float2 bo = *reinterpret_cast<const float2*>(ly.bo + col);
        v0 += bo.x + xs[rr * H + col];
        v1 += bo.y + xs[rr * H + col + 1];
        hst8<true>(io.y, (unsigned)(((n0 + rr) * H + col) * 4), make_uint2(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)));
        const float mean = wave_sum(v0 + v1) / 128.f;
        const float d0 = v0 - mean, d1 = v1 - mean;
        const float m2 = wave_sum(sq_pair(d0, d1));
        if (lane == 0)
            hst8<true>(io.stats_out, (unsigned)(((n0 + rr) * PARTS + pp) * 8), make_uint2(__builtin_bit_cast(unsigned, mean), __builtin_bit_cast(unsigned, m2)));
    }
}

// ---- the step -------------------------------------------------------------------------------------------------
constexpr int LDS_HEAD = 16 + 2 * 48 * 2 * 4 + 2 * NW * 48 * 4;  // abort flag, two sets of row statistics, the reduction scratch
static_assert(LDS_HEAD % 16 == 0, "carve offsets stay 16-byte aligned");
constexpr int LDS_BASES = MAX_SEAMS * 4;

template <typename T, int RG, int HEADS, int PPW>
__global__ __launch_bounds__(NT, 2) void decode_persist_kernel(const PersistArgs a) {
    constexpr int VEC = Vec16<T>::N, H = 64 * VEC, PARTS = H / 128, QW = (2 + HEADS) * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int* abort_flag = reinterpret_cast<int*>(lds);
    float (*rs_x)[2] = reinterpret_cast<float (*)[2]>(lds + 16);
    float (*rs_r)[2] = reinterpret_cast<float (*)[2]>(lds + 16 + 48 * 2 * 4);
    float (*red)[NW][48] = reinterpret_cast<float (*)[NW][48]>(lds + 16 + 2 * 48 * 2 * 4);
    unsigned* bases = reinterpret_cast<unsigned*>(lds + LDS_HEAD);
    unsigned char* work = lds + LDS_HEAD + LDS_BASES;

    const int tid = threadIdx.x;
    // Every phase derives its lane and wave numbers from a copy of the thread index the compiler cannot see through.
    // Otherwise everything a phase computes from the thread index alone (LDS addresses of 48 fragment reads, lane offsets
    // of every load) is loop-invariant in the loop over layers, gets hoisted in front of it and stays live through all
    // four phases of the loop body: hundreds of registers, spilled.
    auto who = [](int& wave, int& lane) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        lane = t & 63;
        wave = __builtin_amdgcn_readfirstlane(t >> 6);
    };
    // workgroups with equal blockIdx % 8 were observed to share an XCD: a cluster is 16 of those
    const int xl = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cluster = (slot / CL) * 8 + xl, j = slot % CL;
    if (cluster >= a.clusters) return;
    const int nseams = 4 * a.L + (a.has_tail ? 1 : 0);

    Seam seam;
    seam.cnt = a.sync + (size_t)cluster * MAX_SEAMS;
    seam.err = a.sync + (size_t)a.clusters * MAX_SEAMS;
    seam.base = bases;
    seam.abort_flag = abort_flag;
    if (tid == 0) *abort_flag = 0;
    // this workgroup has not arrived anywhere yet, so each counter is below its base + 16
    if (tid < nseams) bases[tid] = __hip_atomic_load(seam.cnt + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~15u;
    wg_barrier();
    stamp(a.trace, 0);

    const int N = a.N, R = a.R;
    const int m0 = cluster * CL * R;          // first row of the cluster
    const int b = cluster * CL + j;           // this workgroup's image in the attention phases
    const bool has_image = b < a.B;
    const long long n0 = (long long)b * R;

    const unsigned szT = (unsigned)sizeof(T);
    const hbuf_t r_x{a.x, (unsigned)N * H * 4u}, r_xt{a.xt, (unsigned)N * H * szT};
    const hbuf_t r_y{a.y, (unsigned)N * H * 4u}, r_qkvf{a.qkvf, (unsigned)N * QW * 4u};
    const hbuf_t r_ffh{a.ffh, (unsigned)N * 2u * H * szT}, r_logits{a.logits, (unsigned)N * (unsigned)a.V * 4u};
    auto r_st = [&](int i) { return hbuf_t{i ? a.stats[1] : a.stats[0], (unsigned)N * PARTS * 8u}; };

    bool plain = true;  // the residual stream is x / xt; otherwise LN(y; stats[cs]) with (pg, pb)
    int cs = 0;
    const float *pg = nullptr, *pb = nullptr;
    int seam_idx = 0;   // seams passed so far: phase p waits on seam p - 1 and arrives on seam p
    auto view_x = [&](GemmIO<T>& io) {
        io.x_ln = !plain;
        io.x = r_xt; io.ld_x_b = (unsigned)H * szT;
        io.xraw = r_y; io.xstats = r_st(cs); io.xg = pg; io.xb = pb;
    };
    for (int l = 0; l < a.L; ++l) {
        const PLayer& ly = a.layer[l];
        {   // ---- A: packed in-projection ----
            int wave, lane;
            who(wave, lane);
            GemmIO<T> io{};
            view_x(io);
            io.wf = ly.qkvf_wf; io.bias = ly.qkvf_b;
            io.out = r_qkvf; io.ldc_b = (unsigned)QW * 4u; io.out_plain = a.qkvf; io.out_f32 = true; io.N = QW;
            io.CB = QW / 128; io.shift = 0;
            if (!gemm_phase<T, RG, 16, PARTS, true>(io, seam, seam_idx - 1, j, m0, N, work, rs_x, rs_r, red, wave, lane, a.trace)) return;
            stamp(a.trace, 3 * seam_idx + 2);
            seam_arrive(seam, seam_idx++);
            stamp(a.trace, 3 * seam_idx);
            if (seam_idx - 1 == a.stop_seam) return;
        }
        {   // ---- B: both attentions ----
            const int ns = plain ? 0 : cs ^ 1;
            int wave, lane;
            who(wave, lane);
            if (has_image) {
                AttnPre<T, RG, HEADS, PPW> pre;
                // the ancestry table is the previous launch's: read before the poll, so that the first wave, too, requests
                // its cache rows the moment the seam is passed
                attn_ancestry<T, RG, HEADS, PPW>(pre, a.anc, n0, R, a.t, a.t_cap, wave);
                seam_poll(seam, seam_idx - 1, wave);
                attn_prefetch<T, RG, HEADS, PPW>(pre, ly, a.t, a.t_cap, wave, lane);
                if (!seam_join(seam)) return;
                stamp(a.trace, 3 * seam_idx + 1);
                AttnIO io{};
                io.qkvf = r_qkvf; io.x_plain = plain; io.xr = plain ? r_x : r_y; io.xstats = r_st(cs); io.xg = pg; io.xb = pb;
                io.y = r_y; io.stats_out = r_st(ns);
                io.attn_scale = 1.f;
                if (a.attn_mode == 1) {
                    io.attn_out = a.attn + (long long)a.t * a.S; io.ld_attn = a.t_cap * a.S; io.attn_scale = 1.f / (float)a.L; io.accumulate = l > 0;
                } else if (a.attn_mode == 2) {
                    io.attn_out = a.attn + ((long long)l * a.t_cap + a.t) * N * a.S; io.ld_attn = a.S;
                }
                attn_body<T, RG, HEADS, PPW>(pre, io, ly, b, n0, R, a.S, a.t, a.t_cap, reinterpret_cast<float*>(work), wave, lane);
            }
            plain = false; cs = ns; pg = ly.ln2_g; pb = ly.ln2_b;
            stamp(a.trace, 3 * seam_idx + 2);
            seam_arrive(seam, seam_idx++);
            stamp(a.trace, 3 * seam_idx);
            if (seam_idx - 1 == a.stop_seam) return;
        }
        {   // ---- C: feed-forward 1 ----
            int wave, lane;
            who(wave, lane);
            GemmIO<T> io{};
            view_x(io);
            io.wf = ly.ff1_wf; io.bias = ly.ff1_b;
            io.out = r_ffh; io.ldc_b = 2u * H * szT; io.out_plain = nullptr; io.out_f32 = sizeof(T) == 4; io.N = 2 * H; io.relu = 1;
            io.CB = 2 * H / 128; io.shift = 0;
            if (!gemm_phase<T, RG, 16, PARTS, true>(io, seam, seam_idx - 1, j, m0, N, work, rs_x, rs_r, red, wave, lane, a.trace)) return;
            stamp(a.trace, 3 * seam_idx + 2);
            seam_arrive(seam, seam_idx++);
            stamp(a.trace, 3 * seam_idx);
            if (seam_idx - 1 == a.stop_seam) return;
        }
        {   // ---- D: feed-forward 2 + residual, statistics for the next view ----
            int wave, lane;
            who(wave, lane);
            GemmIO<T> io{};
            io.x_ln = false; io.x = r_ffh; io.ld_x_b = 2u * H * szT;
            io.wf = ly.ff2_wf; io.bias = ly.ff2_b;
            io.r_ln = true; io.rraw = r_y; io.rstats = r_st(cs); io.rg = pg; io.rb = pb;
            io.out = r_y; io.ldc_b = (unsigned)H * 4u; io.out_plain = a.y; io.out_f32 = true; io.N = H;
            io.want_stats = true; io.stats_out = r_st(cs ^ 1);
            io.CB = H / 128; io.shift = 2 * H / 128;  // the workgroups feed-forward 1 left idle take these tiles
            if (!gemm_phase<T, RG, 32, PARTS, true>(io, seam, seam_idx - 1, j, m0, N, work, rs_x, rs_r, red, wave, lane, a.trace)) return;
            cs ^= 1; pg = ly.ln3_g; pb = ly.ln3_b;
            stamp(a.trace, 3 * seam_idx + 2);
            seam_arrive(seam, seam_idx++);
            stamp(a.trace, 3 * seam_idx);
            if (seam_idx - 1 == a.stop_seam) return;
        }
    }
    {   // ---- E: vocabulary projection ----
        int wave, lane;
        who(wave, lane);
        GemmIO<T> io{};
        view_x(io);
        io.wf = a.out_wf; io.bias = a.out_b;
        io.out = r_logits; io.ldc_b = (unsigned)a.V * 4u; io.out_plain = a.logits; io.out_f32 = true; io.N = a.V;
        io.CB = (a.V + 127) / 128; io.shift = 0;
        if (!gemm_phase<T, RG, 16, PARTS, true>(io, seam, seam_idx - 1, j, m0, N, work, rs_x, rs_r, red, wave, lane, a.trace)) return;
        stamp(a.trace, 3 * seam_idx + 2);
    }
    if constexpr (RG >= 2) {
        if (a.has_tail) {   // ---- F: log-softmax + top-K of the image's rows, beam selection, the next position's embedding ----
            seam_arrive(seam, seam_idx++);
            stamp(a.trace, 3 * seam_idx);
            int wave, lane;
            who(wave, lane);
            if (has_image) {
                int t = threadIdx.x;
                asm volatile("" : "+v"(t));
                constexpr int GW = RG <= 2 ? 256 : 128;  // as the stand-alone tail launch: the same sums in the same order
                auto& ts = *reinterpret_cast<TailShared<RG, NT / GW, GW / 64>*>(work);
                beam_tail_body<RG, GW, NT / GW, true>(a.tail, b, t, ts, [&] {
                    seam_poll(seam, seam_idx - 1, wave);
                    const bool ok = seam_join(seam);
                    stamp(a.trace, 120);
                    return ok;
                });
            }
            stamp(a.trace, 121);
        }
    }
}

std::atomic<unsigned long long*> g_trace{nullptr};

template <typename T, int RG, int HEADS> size_t persist_lds_bytes() {
    constexpr int VEC = Vec16<T>::N, H = 64 * VEC;
    const size_t gemm = (size_t)RG * 16 * ((size_t)2 * H * sizeof(T) + 16);
    const size_t attn = sizeof(float) * ((size_t)3 * RG * H + RG * HEADS * 64 + (size_t)NW * RG * H + 3 * H);
    size_t need = LDS_HEAD + LDS_BASES + (gemm > attn ? gemm : attn);
    // more than half of a CU's 160 KiB, so that no two workgroups share a CU (the hand-off is measured one per CU)
    if (need < 84 * 1024) need = 84 * 1024;
    return need;
}

template <typename T, int RG, int HEADS, int PPW> int launch_persist_p(const PersistArgs& a, hipStream_t s) {
    const size_t lds = persist_lds_bytes<T, RG, HEADS>();
    static std::atomic<uint64_t> granted{0};
    MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&decode_persist_kernel<T, RG, HEADS, PPW>), 150 * 1024, granted));
    const int grid = 8 * CL * ((a.clusters + 7) / 8);
    MIC_LAUNCH((decode_persist_kernel<T, RG, HEADS, PPW>), dim3(grid), dim3(NT), lds, s, a);
    return 0;
}
template <typename T, int RG, int HEADS> int launch_persist(const PersistArgs& a, hipStream_t s) {
    return a.t_cap <= 32 ? launch_persist_p<T, RG, HEADS, 4>(a, s) : launch_persist_p<T, RG, HEADS, 6>(a, s);
}

template <typename T, int HEADS> int dispatch_persist(const PersistArgs& a, hipStream_t s) {
    switch (a.R) {
        case 1: return launch_persist<T, 1, HEADS>(a, s);
        case 2: return launch_persist<T, 2, HEADS>(a, s);
        case 3: return launch_persist<T, 3, HEADS>(a, s);
        default: mic_set_error("mic_tr_step_persistent: %d rows per image", a.R); return MIC_ERR_ARG;
    }
}

}  // namespace

extern "C" int mic_tr_persist_supported(int dtype, int H, int heads, int layers, int S, int rows_per_image, int t_cap, int N) {
    if (dtype != MIC_F32 && dtype != MIC_BF16) return 0;
    const int vec = 16 / (int)mic_dtype_size(dtype);
    if (H != 64 * vec || (heads != 1 && heads != 2) || S < 1 || S > 64 || rows_per_image < 1 || rows_per_image > 3) return 0;
    if (t_cap < 1 || t_cap > 48 || layers < 1 || layers > MIC_MAX_LAYERS || 4 * layers > MAX_SEAMS) return 0;
    if (N < 1 || N % rows_per_image != 0) return 0;
    const int B = N / rows_per_image;
    return (B + CL - 1) / CL <= 16;  // one workgroup per CU: at most 16 clusters of 16 on the 256 CUs
}

extern "C" int mic_tr_persist_trace(void* buffer) {
    g_trace.store(static_cast<unsigned long long*>(buffer));
    return 0;
}

extern "C" int64_t mic_tr_persist_sync_words(int N, int rows_per_image) {
    if (N < 1 || rows_per_image < 1) return -1;
    const int B = N / rows_per_image;
    return (int64_t)((B + CL - 1) / CL) * MAX_SEAMS + 64;
}

// Whether (dec, ws) can take the persistent step: the fused-attention conditions, a supported shape and the seam counters.
bool mic_tr_persist_eligible(const mic_tr_decoder* dec, const mic_tr_workspace* ws) {
    static const bool off = getenv("MIC_NO_PERSIST") != nullptr;
    if (off || !ws->sync || !ws->qkvf || !mic_tr_fused_eligible(dec, ws)) return false;
    return mic_tr_persist_supported(dec->dtype, dec->H, dec->heads, dec->layers, ws->S, ws->rows_per_image, ws->t_cap, ws->N) != 0 &&
           dec->vocab % 4 == 0 && (long long)ws->N * dec->vocab * 4 < (1ll << 31);
}

// Whether the beam-search tail of a step (K beams over V words) can run as the last phase of the one-launch step: two
// (K = 2) or four (K = 3) groups of threads hold a row of up to 4096 logits each, one seam more.
bool mic_tr_persist_tail_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws, int K, int V) {
    static const bool off = getenv("MIC_NO_PERSIST_TAIL") != nullptr;
    return !off && K >= 2 && K == ws->rows_per_image && K <= 3 && V == dec->vocab && V <= 4096 && 4 * dec->layers + 1 <= MAX_SEAMS && dec->H % 4 == 0;
}

// The layers and the vocabulary projection of one step on the residual stream already in ws->x / ws->xt (the embedding
// is the caller's: mic_embed_pos, or the beam tail of the previous step).
int mic_tr_persist_launch(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* anc, int t, int attn_mode,
                          const mic_beam_tail_desc* tail, hipStream_t s) {
    PersistArgs a{};
    if (tail) {
        MIC_ARG_CHECK(mic_tr_persist_tail_ok(dec, ws, tail->K, tail->V) && tail->B * tail->K == ws->N && tail->logits == ws->logits &&
                          tail->ld == dec->vocab && tail->n_gather == 0 && (tail->K <= 2 || tail->row_group_threads == 128),
                      "mic_tr_step: the beam tail does not fit the one-launch step");
        MIC_TRY(tail_args_from_desc(tail, a.tail));
        a.has_tail = 1;
    }
    for (int l = 0; l < dec->layers; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        PLayer& p = a.layer[l];
        p.qkvf_wf = Y.self_qkvf_wf; p.qkvf_b = Y.self_qkvf_b;
        p.kcache = ws->kcache[l]; p.vfcache = ws->vfcache[l];
        p.self_b = Y.self_of_b; p.ln1_g = Y.ln1_g; p.ln1_b = Y.ln1_b;
        p.kf = ws->cross_kf[l]; p.vf = ws->cross_vf[l]; p.c = ws->cross_c[l]; p.bo = Y.cross_out_b;
        p.ln2_g = Y.ln2_g; p.ln2_b = Y.ln2_b;
        p.ff1_wf = Y.ff1_wf; p.ff1_b = Y.ff1_b; p.ff2_wf = Y.ff2_wf; p.ff2_b = Y.ff2_b;
        p.ln3_g = Y.ln3_g; p.ln3_b = Y.ln3_b;
    }
    a.out_wf = dec->out_wf; a.out_b = dec->out_b;
    a.x = ws->x; a.xt = ws->xt; a.y = ws->y; a.qkvf = ws->qkvf; a.ffh = ws->ffh; a.logits = ws->logits;
    a.stats[0] = ws->stats[0]; a.stats[1] = ws->stats[1];
    a.anc = anc; a.attn = ws->attn; a.sync = ws->sync;
    a.N = ws->N; a.R = ws->rows_per_image; a.B = ws->N / ws->rows_per_image; a.S = ws->S; a.t = t; a.t_cap = ws->t_cap;
    a.L = dec->layers; a.V = dec->vocab; a.attn_mode = attn_mode;
    a.clusters = (a.B + CL - 1) / CL;
    {
        const char* e = getenv("MIC_PERSIST_STOP");  // diagnostic runs only
        a.stop_seam = e ? atoi(e) : -1;
    }
    a.trace = g_trace.load();
    if (dec->dtype == MIC_BF16) return dec->heads == 2 ? dispatch_persist<bf16_t, 2>(a, s) : dispatch_persist<bf16_t, 1>(a, s);
    return dec->heads == 2 ? dispatch_persist<float, 2>(a, s) : dispatch_persist<float, 1>(a, s);
}
