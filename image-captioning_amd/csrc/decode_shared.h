// Device helpers shared by the decode-step kernels: the one-launch-per-sub-layer kernels (gemm_ln.hip, attn_fused.hip)
// and the persistent step (decode_persist.hip) call the same functions for every piece of arithmetic whose rounding
// depends on how the compiler contracts it, so the two schedules produce the same bits from the same inputs.
#pragma once
#include "mic_common.h"

// One step of the pairwise mean / centred-sum-of-squares update of Chan et al.: folds a part of nb values with mean mb
// and centred sum of squares sb into the running (n, mu, m2).
__device__ __forceinline__ void chan_step(float& n, float& mu, float& m2, float nb, float mb, float sb) {
#pragma clang fp contract(off)
    const float nt = n + nb, d = mb - mu;
    mu = mu + d * (nb / nt);
    m2 = m2 + (sb + (d * d) * (n * nb / nt));
    n = nt;
}
__device__ __forceinline__ float chan_rstd(float m2, float n) {
#pragma clang fp contract(off)
    return 1.0f / sqrtf(m2 / n + 1e-5f);
}
// LayerNorm of an already centred value
__device__ __forceinline__ float ln_scale(float centred, float rstd, float g, float b) {
#pragma clang fp contract(off)
    return __builtin_fmaf(centred * rstd, g, b);
}

// d0^2 + d1^2 and a * b + c with the contraction spelled out
__device__ __forceinline__ float sq_pair(float d0, float d1) {
#pragma clang fp contract(off)
    return __builtin_fmaf(d0, d0, d1 * d1);
}

// v of the lane whose number differs in bit O.  Within a row of 16 lanes this is a data-parallel-primitive move of the
// vector ALU (quad permutes for 1 and 2, a rotation by 8 within the row for 8); the others go through the LDS crossbar.
// Exchanges across rows of 16 lanes on the vector ALU as well: v_permlane16_swap trades the odd rows of one register with
// the even rows of another, v_permlane32_swap the upper half of one with the lower half of another; with the same value in
// both, one result holds every pair's even-row (lower-half) value and the other its odd-row (upper-half) value.
template <int O> __device__ __forceinline__ float lane_xor(float v) {
    if constexpr (O == 1) return MIC_DPP_F32(v, 0xB1);        // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return MIC_DPP_F32(v, 0x4E);   // quad_perm [2,3,0,1]
    else if constexpr (O == 8) return MIC_DPP_F32(v, 0x128);  // row_ror:8
    else if constexpr (O == 16 || O == 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if constexpr (O == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            return __builtin_bit_cast(float, (lane & 16u) ? r[0] : r[1]);
        } else {
            const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            return __builtin_bit_cast(float, (lane & 32u) ? r[0] : r[1]);
        }
    } else return __shfl_xor(v, O, 64);
}
// v + the value of the lane 16 (32) away.  Written through lane_xor: adding the two results of one swap of a register
// with itself comes out wrong with hipcc 7.2 (2 * v; tools/permlane_check.hip), selecting one of them does not.
__device__ __forceinline__ float sum_xor16(float v) { return v + lane_xor<16>(v); }
__device__ __forceinline__ float sum_xor32(float v) { return v + lane_xor<32>(v); }
// v summed over the lanes of an aligned group of W lanes (W a power of two up to 64); every lane of the group returns the sum.
// The order of the additions is that of a butterfly from the widest exchange down.
template <int W> __device__ __forceinline__ float group_sum_down(float v) {
    if constexpr (W >= 64) v += lane_xor<32>(v);
    if constexpr (W >= 32) v += lane_xor<16>(v);
    if constexpr (W >= 16) v += lane_xor<8>(v);
    if constexpr (W >= 8) v += lane_xor<4>(v);
    if constexpr (W >= 4) v += lane_xor<2>(v);
    if constexpr (W >= 2) v += lane_xor<1>(v);
    return v;
}

// One halving step over lane bit O: a lane keeps HALF of its 2 * HALF values and hands the other half to its partner.
template <int HALF, int O, int N> __device__ __forceinline__ int halve(float (&v)[N], int lane) {
    const bool up = (lane & O) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        // the two values are made opaque first: otherwise the selects below are turned into one register-array access
        // with a runtime index, which is lowered to a compare-and-select chain over all N registers
        float lo = v[i], hi = v[i + HALF];
        asm volatile("" : "+v"(lo), "+v"(hi));
        const float send = up ? lo : hi;
        const float keep = up ? hi : lo;
        v[i] = keep + lane_xor<O>(send);
    }
    return up ? HALF : 0;
}

// Sums N values (a power of two, N <= WIDTH) over groups of WIDTH lanes by recursive halving, so that N - 1
// (+ log2(WIDTH / N)) cross-lane moves do the work of N * log2(WIDTH).  The lane bits are used in the order 1, 2, 8, 4,
// 16, 32: the steps that move the most values are the cheap in-row ones.  On return v[0] holds the group's total of the
// value whose index is returned; `writer` is true on one lane of those that hold the same total.
template <int N, int WIDTH> __device__ __forceinline__ int reduce_scatter(float (&v)[N], int lane, bool& writer) {
    static_assert(N >= 1 && (N & (N - 1)) == 0 && N <= WIDTH && (WIDTH == 32 || WIDTH == 64), "reduce_scatter sizes");
    int idx = 0;
    if constexpr (N >= 2) idx += halve<N / 2, 1>(v, lane);
    if constexpr (N >= 4) idx += halve<N / 4, 2>(v, lane);
    if constexpr (N >= 8) idx += halve<N / 8, 8>(v, lane);
    if constexpr (N >= 16) idx += halve<N / 16, 4>(v, lane);
    if constexpr (N >= 32) idx += halve<N / 32, 16>(v, lane);
    if constexpr (N >= 64) idx += halve<N / 64, 32>(v, lane);
    constexpr int used = N >= 64 ? 63 : N >= 32 ? 31 : N >= 16 ? 15 : N >= 8 ? 11 : N >= 4 ? 3 : N >= 2 ? 1 : 0;
    if constexpr (!(used & 1)) v[0] += lane_xor<1>(v[0]);
    if constexpr (!(used & 2)) v[0] += lane_xor<2>(v[0]);
    if constexpr (!(used & 4)) v[0] += lane_xor<4>(v[0]);
    if constexpr (!(used & 8)) v[0] += lane_xor<8>(v[0]);
    if constexpr (!(used & 16)) v[0] += lane_xor<16>(v[0]);
    if constexpr (!(used & 32) && WIDTH == 64) v[0] += lane_xor<32>(v[0]);
    writer = (lane & (WIDTH - 1) & ~used) == 0;
    return idx;
}

constexpr int pow2_at_least(int n) { return n <= 1 ? 1 : 2 * pow2_at_least((n + 1) / 2); }

// Softmax of one row of up to 64 scores held one per lane (lanes at and beyond `count` hold nothing): returns this
// lane's probability.
__device__ __forceinline__ float wave_softmax(float v, bool on) {
    const float m = wave_max(on ? v : -INFINITY);
    const float e = on ? expf(v - m) : 0.f;
    const float tot = wave_sum(e);
    return e / tot;
}
