// Both attention sub-layers of a post-LN decoder layer for one new position, one launch, one workgroup per image:
//
//   y1 = x + sum_h softmax_p( q_h . k_h[p] / sqrt(d) ) V'_h[p] + b'          self-attention over the per-beam cache
//   u  = LayerNorm1(y1)
//   y2 = u + sum_h softmax_s( u . K'_h[s] + c_h[s] ) V'x_h[s] + b_o           cross-attention over the image tokens
//
// (+ the per-128-column row statistics of y2, from which the next launch applies LayerNorm2 while loading).
// Both output projections are folded into the values: V'_h[p] = W_o[:, head h] v_h[p] is what the cache holds
// (H elements per head and position instead of H / heads), so the attention-weighted sum IS the projected
// output and no GEMM launch sits between the two attentions; b' = b_o + sum_h W_o[:, head h] b_v,h because the
// softmax weights sum to one.  The cross-attention side is the folded form of mic_cross_attn_folded.
// Layer by layer a step needed self-attention, output projection, and folded cross-attention launches, each
// a chain of dependent memory round trips behind a kernel boundary; here the rows of an image stay in LDS.
//
// 8 waves.  Self-attention: wave w owns the (row, position) pairs w, w+8, ... of up to RG rows x 32 positions;
// a key row is one coalesced 16-byte-per-lane wave load and a folded value row `heads` of them, all requested
// up front through the ancestry table.  Cross-attention: wave w owns image tokens w, w+8, ...; its folded keys
// are requested as soon as the self-attention keys have been consumed and its folded values as soon as the
// self-attention values have, so both are in flight during the softmax / LayerNorm in between.
#include "mic_common.h"
#include "decode_shared.h"

#include <atomic>

#include <stdlib.h>

namespace {

struct FusedAttnArgs {
    const float* qkvf;     // [N][(2+heads)H]: q | k | v'_0 | ... of the new position
    void* kcache;          // [N][t_cap][H]
    void* vfcache;         // [N][t_cap][heads][H]
    const int32_t* anc;    // [N][t_cap] or null
    const float* x_plain;  // [N][H] residual stream entering the layer, or null: then LN(raw; stats) * gamma + beta
    const float *raw, *stats, *gamma, *beta;
    const float* self_b;   // [H] b'
    const float *ln1_g, *ln1_b;
    const void *kf, *vf;   // [B][S][heads][H]
    const float* c;        // [B][S][heads]
    const float* bo;       // [H]
    float* y;              // [N][H] (may alias raw)
    float* stats_out;      // [N][H/128][2]
    float* attn_out;
    int ld_attn;
    float attn_scale;
    int accumulate, t, t_cap, S, R, N;
};

// NG: row groups of an image (more than one when it has more rows than a group takes); they run one after another inside
// the workgroup, the folded cross keys / values re-requested per group (from L2 after the first).  The group count is a
// compile-time constant and the loop over groups fully unrolled: a real loop makes the compiler hoist every
// group-independent value out of it and keep it alive across the whole kernel, which spills.
// PPW: cached positions per wave and row, 8 * PPW positions over the waves (4: up to 32 positions, 6: up to 48).  With 6 the
// folded values of a wave's last three positions are requested only once its keys have been consumed (their registers).
template <typename T, int RG, int HEADS, int NG, int PPW = 4>
__global__ __launch_bounds__(512) void attn_fused_kernel(const FusedAttnArgs a) {
    constexpr int VEC = Vec16<T>::N, H = 64 * VEC, PARTS = H / 128, SP = 64, NW = 8, NT = 64 * NW;
    constexpr int PPE = PPW == 4 ? 4 : 3, PPL = PPW - PPE;  // positions whose values are requested up front / late
    constexpr int NJ = RG * PPW, NJP = pow2_at_least(NJ);  // (row, position) pairs per wave
    constexpr int NI = 64 / NW, NX = NI * HEADS;  // image tokens per wave: up to 64 over the waves
    constexpr int F4 = H / 4, NRV = (RG * F4 + NT - 1) / NT;
    constexpr int QW = (2 + HEADS) * H, LPH = 64 / HEADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                       // [RG][H] residual stream entering the layer
    float* xs = xin + RG * H;                // [RG][H] LayerNorm1(y1)
    float* qs = xs + RG * H;                 // [RG][H] scaled queries
    float* sc = qs + RG * H;                 // [RG][HEADS][SP]
    float* part = sc + RG * HEADS * SP;      // [NW][RG][H]
    float* vecs = part + NW * RG * H;        // [3][H]: folded output bias of the self-attention, LayerNorm1 weight and bias
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform values are kept in scalar registers: the cache rows of a wave then differ from one another by a scalar
    // offset only, and all of its loads share one vector register of lane offsets
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, R = a.R, S = a.S, t = a.t, P = t + 1;
    const int lc = lane * VEC, hl = lane / LPH;
    const float scale = sqrtf(1.0f / (float)(H / HEADS));
    const T* kc = (const T*)a.kcache;
    const T* vc = (const T*)a.vfcache;
    // addresses are written as (wave-uniform base + wave-uniform offset) + lane offset, so the uniform part stays in scalar registers
    const T* kfb = (const T*)a.kf + (long long)b * S * HEADS * H;
    const T* vfb = (const T*)a.vf + (long long)b * S * HEADS * H;

    if (tid < 3 * F4) {  // published by the barrier every row group passes before its first use
        const int which = tid / F4, c4 = (tid - which * F4) * 4;
        const float* src = which == 0 ? a.self_b : which == 1 ? a.ln1_g : a.ln1_b;
        *reinterpret_cast<float4*>(vecs + which * H + c4) = *reinterpret_cast<const float4*>(src + c4);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int r0 = g * RG;
        const int rn = NG == 1 ? R : R - r0 < RG ? R - r0 : RG;
        const long long n0 = (long long)b * R + r0;

        // ---- every load that depends on nothing is requested first ----
        // residual rows and queries (one float4 per thread and round), with the statistics of the row when it is a view
        float4 rv[NRV], gv[NRV], bv[NRV], qv[NRV];
        float2 stp[NRV][PARTS];
#pragma unroll
        for (int u = 0; u < NRV; ++u) {
            const int i = tid + NT * u, rr = i < rn * F4 ? i / F4 : 0, c4 = (i - (i / F4) * F4) * 4;
            const float* src = a.x_plain ? a.x_plain : a.raw;
            rv[u] = *reinterpret_cast<const float4*>(src + (n0 + rr) * H + c4);
            qv[u] = *reinterpret_cast<const float4*>(a.qkvf + (n0 + rr) * QW + c4);
            if (!a.x_plain) {
                gv[u] = *reinterpret_cast<const float4*>(a.gamma + c4);
                bv[u] = *reinterpret_cast<const float4*>(a.beta + c4);
#pragma unroll
                for (int p = 0; p < PARTS; ++p) stp[u][p] = *reinterpret_cast<const float2*>(a.stats + ((n0 + rr) * PARTS + p) * 2);
            }
        }
        // Cache rows of this wave's pairs: ancestry (one round trip), then keys and folded values (a second).  Every load
        // is unconditional, from a clamped (always valid) cell, and masked when used: a load under a condition is
        // branched around and waited for at once, which would serialise the round trips.
        int srow[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int rr = j / PPW, p = wave + NW * (j % PPW);
            const int own = (int)n0 + (rr < rn ? rr : 0);
            srow[j] = own;
            if (a.anc) {
                const int got = __builtin_amdgcn_readfirstlane(a.anc[(long long)own * a.t_cap + (p < a.t_cap ? p : 0)]);
                srow[j] = p < t ? got : own;  // entries at and beyond t are not ancestry yet
            }
        }
        Vec16<T> kreg[NJ], vreg[RG * PPE][HEADS];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = j % PPW, p = wave + NW * q;
            const long long cell = (long long)srow[j] * a.t_cap + (p < t ? p : 0);
            kreg[j] = ld16(kc + cell * H + lc);
            if (q < PPE) {
#pragma unroll
                for (int h = 0; h < HEADS; ++h) vreg[(j / PPW) * PPE + q][h] = ld16(vc + (cell * HEADS + h) * H + lc);
            }
        }
        // the new position of row w, by wave w: the fresh projection rounded through T, exactly what the cache will hold
        Vec16<T> knew, vnew[HEADS];
        if (wave < rn) {
            const float* f = a.qkvf + (n0 + wave) * QW + H + lc;
            const long long cell = (n0 + wave) * a.t_cap + t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) knew.v[e] = from_f32<T>(f[e]);
            st16((T*)a.kcache + cell * H + lc, knew);
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) vnew[h].v[e] = from_f32<T>(f[(1 + h) * H + e]);
                st16((T*)a.vfcache + (cell * HEADS + h) * H + lc, vnew[h]);
            }
        }

        // ---- residual stream and scaled queries of the group's rows -> LDS ----
#pragma unroll
        for (int u = 0; u < NRV; ++u) {
            const int i = tid + NT * u, rr = i / F4, c4 = (i - rr * F4) * 4;
            if (i < rn * F4) {
                float4 v = rv[u];
                if (!a.x_plain) {
                    // mean and 1/sqrt(var + eps) of the row from its per-part (mean, M2) pairs (pairwise update of Chan et al.)
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
        __syncthreads();

        // ---- self-attention scores ----
        {
            float pv[NJP];
#pragma unroll
            for (int j = 0; j < NJP; ++j) {
                pv[j] = 0.f;
                if (j < NJ) {
                    const float* qp = qs + (j / PPW) * H + lc;
#pragma unroll
                    for (int e4 = 0; e4 < VEC; e4 += 4) {
                        const float4 qq = *reinterpret_cast<const float4*>(qp + e4);
                        pv[j] = fmaf(qq.x, to_f32<T>(kreg[j].v[e4]), pv[j]);
                        pv[j] = fmaf(qq.y, to_f32<T>(kreg[j].v[e4 + 1]), pv[j]);
                        pv[j] = fmaf(qq.z, to_f32<T>(kreg[j].v[e4 + 2]), pv[j]);
                        pv[j] = fmaf(qq.w, to_f32<T>(kreg[j].v[e4 + 3]), pv[j]);
                    }
                }
            }
            bool writer;
            const int j = reduce_scatter<NJP, LPH>(pv, lane, writer);
            const int rr = j / PPW, p = wave + NW * (j % PPW);
            if (writer && j < NJ && rr < rn && p < t) sc[(rr * HEADS + hl) * SP + p] = pv[0];
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
        for (int j = 0; j < NJ; ++j) {
            const int q = j % PPW, p = wave + NW * q;
            if (q >= PPE) {
                const long long cell = (long long)srow[j] * a.t_cap + (p < t ? p : 0);
#pragma unroll
                for (int h = 0; h < HEADS; ++h) vlate[(j / PPW) * PPL + q - PPE][h] = ld16(vc + (cell * HEADS + h) * H + lc);
            }
        }
        Vec16<T> kx[NI][HEADS];
        float cx[NI][HEADS];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + NW * i, pc = p < S ? p : S - 1;  // tokens past S: a valid row, masked when used
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                kx[i][h] = ld16(kfb + (pc * HEADS + h) * H + lc);
                cx[i][h] = a.c[((long long)b * S + pc) * HEADS + h];
            }
        }
        __syncthreads();
        for (int rh = wave; rh < rn * HEADS; rh += NW) {
            float* row = sc + rh * SP;
            const float pr = wave_softmax(lane < P ? row[lane] : 0.f, lane < P);
            if (lane < P) row[lane] = pr;
        }
        __syncthreads();
        // ---- self-attention output, one row at a time: this wave's share -> LDS ----
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
            for (int jj = 0; jj < PPW; ++jj) {
                const int j = rr * PPW + jj, p = wave + NW * jj;
                const bool on = rr < rn && p < t;
#pragma unroll
                for (int h = 0; h < HEADS; ++h) {
                    // a masked pair was loaded from a cell nothing has written yet: its bytes are selected away, not
                    // multiplied by a zero weight (0 * NaN is NaN, and a cache need not be zero-filled)
                    const Vec16<T>& held = jj < PPE ? vreg[rr * PPE + jj][h] : vlate[PPL > 0 ? rr * PPL + jj - PPE : 0][h];
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
        // folded cross values, requested once the self-attention values have been consumed: in flight through LayerNorm1,
        // the cross scores and their softmax
        Vec16<T> vx[NI][HEADS];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + NW * i, pc = p < S ? p : S - 1;
#pragma unroll
            for (int h = 0; h < HEADS; ++h) vx[i][h] = ld16(vfb + (pc * HEADS + h) * H + lc);
        }
        __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
        for (int rh = wave; rh < rn * HEADS; rh += NW) {
            float* row = sc + rh * SP;
            const float pr = wave_softmax(lane < S ? row[lane] : 0.f, lane < S);
            if (lane < S) row[lane] = pr;
        }
        __syncthreads();
        if (a.attn_out) {
            const float hs = a.attn_scale / (float)HEADS;
            for (int i = tid; i < rn * S; i += NT) {
                const int rr = i / S, p = i - rr * S;
                float sum = 0.f;
#pragma unroll
                for (int hh = 0; hh < HEADS; ++hh) sum += sc[(rr * HEADS + hh) * SP + p];
                float* dst = a.attn_out + (n0 + rr) * a.ld_attn + p;
                *dst = __builtin_fmaf(sum, hs, a.accumulate ? *dst : 0.f);
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
        __syncthreads();
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
            const float2 bo = *reinterpret_cast<const float2*>(a.bo + col);
            v0 += bo.x + xs[rr * H + col];
            v1 += bo.y + xs[rr * H + col + 1];
            *reinterpret_cast<float2*>(a.y + (n0 + rr) * H + col) = make_float2(v0, v1);
            const float mean = wave_sum(v0 + v1) / 128.f;
            const float d0 = v0 - mean, d1 = v1 - mean;
            const float m2 = wave_sum(sq_pair(d0, d1));
            if (lane == 0 && a.stats_out) {
                float* dst = a.stats_out + ((n0 + rr) * PARTS + pp) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
        }
        __syncthreads();
    }
}

template <typename T, int RG, int HEADS, int NG, int PPW = 4> int launch_fused_attn_p(const FusedAttnArgs& a, hipStream_t s) {
    constexpr int H = 64 * Vec16<T>::N;
    const size_t lds = sizeof(float) * ((size_t)3 * RG * H + RG * HEADS * 64 + 8 * RG * (size_t)H + 3 * H);
    static std::atomic<uint64_t> granted{0};  // one bit per device; lanes call from several host threads
    MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&attn_fused_kernel<T, RG, HEADS, NG, PPW>), 150 * 1024, granted));
    MIC_LAUNCH((attn_fused_kernel<T, RG, HEADS, NG, PPW>), dim3(a.N / a.R), dim3(512), lds, s, a);
    return 0;
}
template <typename T, int RG, int HEADS, int NG> int launch_fused_attn(const FusedAttnArgs& a, hipStream_t s) {
    return a.t_cap <= 32 ? launch_fused_attn_p<T, RG, HEADS, NG, 4>(a, s) : launch_fused_attn_p<T, RG, HEADS, NG, 6>(a, s);
}

template <typename T, int HEADS> int dispatch_fused_attn(const FusedAttnArgs& a, hipStream_t s) {
    if (a.R == 1) return launch_fused_attn<T, 1, HEADS, 1>(a, s);
    if (a.R == 2) return launch_fused_attn<T, 2, HEADS, 1>(a, s);
    if (a.R == 3) return launch_fused_attn<T, 3, HEADS, 1>(a, s);
    if (a.R == 4) return launch_fused_attn<T, 2, HEADS, 2>(a, s);
    if (a.R <= 6) return launch_fused_attn<T, 3, HEADS, 2>(a, s);
    return launch_fused_attn<T, 3, HEADS, 3>(a, s);
}

}  // namespace

extern "C" int mic_attn_fused_supported(int dtype, int H, int heads, int S, int rows_per_image, int t_cap) {
    if (dtype != MIC_F32 && dtype != MIC_BF16) return 0;
    const int vec = 16 / (int)mic_dtype_size(dtype);
    return H == 64 * vec && (heads == 1 || heads == 2) && S >= 1 && S <= 64 && rows_per_image >= 1 && rows_per_image <= MIC_MAX_BEAM &&
           t_cap >= 1 && t_cap <= 48;
}

extern "C" int mic_attn_fused(const mic_attn_fused_desc* d, void* stream) {
    MIC_ARG_CHECK(d && d->qkvf && d->kcache && d->vfcache && d->self_b && d->ln1_g && d->ln1_b && d->kf && d->vf && d->c && d->bo && d->y,
                  "mic_attn_fused: null pointer");
    MIC_ARG_CHECK(mic_attn_fused_supported(d->dtype, d->H, d->heads, d->S, d->rows_per_image, d->t_cap),
                  "mic_attn_fused: unsupported shape (dtype %d, H=%d, heads=%d, S=%d, rows per image %d, t_cap=%d)", d->dtype, d->H, d->heads,
                  d->S, d->rows_per_image, d->t_cap);
    MIC_ARG_CHECK(d->N > 0 && d->N % d->rows_per_image == 0, "mic_attn_fused: N=%d is not a multiple of rows_per_image=%d", d->N,
                  d->rows_per_image);
    MIC_ARG_CHECK(d->t >= 0 && d->t < d->t_cap, "mic_attn_fused: position %d outside t_cap=%d", d->t, d->t_cap);
    MIC_ARG_CHECK((d->x != nullptr) != (d->x_ln.raw != nullptr), "mic_attn_fused: exactly one of x and x_ln must be given");
    if (d->x_ln.raw) {
        MIC_ARG_CHECK(d->x_ln.stats && d->x_ln.gamma && d->x_ln.beta && d->x_ln.ld == d->H && d->x_ln.width == d->H && d->x_ln.parts == d->H / 128,
                      "mic_attn_fused: LayerNorm view must cover dense rows of H=%d columns in %d parts", d->H, d->H / 128);
    }
    MIC_ARG_CHECK(!d->attn_out || d->ld_attn >= d->S, "mic_attn_fused: ld_attn=%d < S=%d", d->ld_attn, d->S);
    const uintptr_t al = (uintptr_t)d->qkvf | (uintptr_t)d->kcache | (uintptr_t)d->vfcache | (uintptr_t)d->x | (uintptr_t)d->x_ln.raw |
                         (uintptr_t)d->x_ln.gamma | (uintptr_t)d->x_ln.beta | (uintptr_t)d->self_b | (uintptr_t)d->ln1_g | (uintptr_t)d->ln1_b |
                         (uintptr_t)d->kf | (uintptr_t)d->vf | (uintptr_t)d->bo | (uintptr_t)d->y;
    MIC_ARG_CHECK((al & 15) == 0 && ((uintptr_t)d->x_ln.stats & 7) == 0 && ((uintptr_t)d->stats_out & 7) == 0,
                  "mic_attn_fused: pointers must be 16-byte aligned (statistics 8-byte)");
    FusedAttnArgs a{};
    a.qkvf = d->qkvf; a.kcache = d->kcache; a.vfcache = d->vfcache; a.anc = d->anc;
    a.x_plain = d->x; a.raw = d->x_ln.raw; a.stats = d->x_ln.stats; a.gamma = d->x_ln.gamma; a.beta = d->x_ln.beta;
    a.self_b = d->self_b; a.ln1_g = d->ln1_g; a.ln1_b = d->ln1_b;
    a.kf = d->kf; a.vf = d->vf; a.c = d->c; a.bo = d->bo; a.y = d->y; a.stats_out = d->stats_out;
    a.attn_out = d->attn_out; a.ld_attn = d->ld_attn; a.attn_scale = d->attn_scale; a.accumulate = d->accumulate;
    a.t = d->t; a.t_cap = d->t_cap; a.S = d->S; a.R = d->rows_per_image; a.N = d->N;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MIC_BF16) return d->heads == 2 ? dispatch_fused_attn<bf16_t, 2>(a, s) : dispatch_fused_attn<bf16_t, 1>(a, s);
    return d->heads == 2 ? dispatch_fused_attn<float, 2>(a, s) : dispatch_fused_attn<float, 1>(a, s);
}
