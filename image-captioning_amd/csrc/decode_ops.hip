// Per-step decoder kernels that are not GEMMs: embedding, LayerNorm, single-query attention over
// the per-beam KV cache / per-image cross K/V, LSTM cell, beam-state row gather.
// These are HBM/L2-bandwidth and latency bound (one new token per row); every global access is a
// 16-byte (or 4-float) vector along the hidden axis.
#include "mic_common.h"

#include <atomic>

#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_pos_kernel(const int32_t* __restrict__ ids, int ld_ids, int pos, const float* __restrict__ tok_emb,
                                 const float* __restrict__ pos_emb, float* __restrict__ x, T* __restrict__ xt, int ld_xt,
                                 int N, int H, int vocab) {
    const int hv = H >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * hv) return;
    const int n = i / hv, c = (i - n * hv) << 2;
    int id = ids[(long long)n * ld_ids + pos];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never index outside the table
    float4 v = *reinterpret_cast<const float4*>(tok_emb + (long long)id * H + c);
    if (pos_emb) {
        const float4 p = *reinterpret_cast<const float4*>(pos_emb + (long long)pos * H + c);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (x) *reinterpret_cast<float4*>(x + (long long)n * H + c) = v;
    if (xt) {
        T* o = xt + (long long)n * ld_xt + c;
        o[0] = from_f32<T>(v.x); o[1] = from_f32<T>(v.y); o[2] = from_f32<T>(v.z); o[3] = from_f32<T>(v.w);
    }
}

// ---------------------------------------------------------------------------------------------
// one wave per row, 4 rows per block
template <typename T>
__global__ void layernorm_kernel(const float* x, const float* __restrict__ g, const float* __restrict__ b,
                                 const float* res, float* y, T* __restrict__ yt, int ld_yt, int N, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (long long)row * H;
    float s = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        s += ln_sum4(v.x, v.y, v.z, v.w);
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        q += ln_sq4(v.x, v.y, v.z, v.w, mean);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        const float4 gg = *reinterpret_cast<const float4*>(g + i), bb = *reinterpret_cast<const float4*>(b + i);
        float4 o;
        o.x = ln_apply(v.x, mean, rstd, gg.x, bb.x);
        o.y = ln_apply(v.y, mean, rstd, gg.y, bb.y);
        o.z = ln_apply(v.z, mean, rstd, gg.z, bb.z);
        o.w = ln_apply(v.w, mean, rstd, gg.w, bb.w);
        if (res) {  // residual-post-norm: the normalised branch is added to the stream (res may alias y)
            const float4 r = *reinterpret_cast<const float4*>(res + (long long)row * H + i);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (y) *reinterpret_cast<float4*>(y + (long long)row * H + i) = o;
        if (yt) {
            T* t = yt + (long long)row * ld_yt + i;
            t[0] = from_f32<T>(o.x); t[1] = from_f32<T>(o.y); t[2] = from_f32<T>(o.z); t[3] = from_f32<T>(o.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-query attention: one workgroup (256 threads) per row, all heads.
//   phase A: score[h][p] = q_h . k_p      (16-lane group per (h,p) pair, 16 B per lane per load)
//   phase B: softmax over p per head      (one wave per head)
//   phase C: o = sum_p w[h][p] * v_p      (threads = 16-byte channel chunks x position groups)
struct AttnArgs {
    const float* q;   // self: qkv [N][3H]; cross: q [N][ldq]
    int ldq;
    void* kcache;     // self: [N][t_cap][H]
    void* vcache;
    const int32_t* anc;
    const void* kv;   // cross: [B][S][2H]
    int t, t_cap, S, rows_per_image;
    void* o;
    int ld_o;
    float* attn_out;
    int ld_attn;
    float attn_scale;
    int accumulate;
    int N, H, heads;
};

// CPL = 16-byte chunks of one head's key row handled by one lane (head_dim / VEC / 16, rounded up).
// Loads are issued in rounds of up to 8 per lane before anything consumes them: the value rows of
// phase C do not depend on the softmax, so they are requested together with the first key rows and
// the kernel pays about two dependent memory round trips instead of five.
template <typename T, bool SELF, int CPL>
__global__ __launch_bounds__(256) void attn_step_kernel(const AttnArgs a) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int DEPTH = 8;                  // 16-byte loads in flight per lane per round
    constexpr int RA = DEPTH / CPL > 0 ? DEPTH / CPL : 1;  // (head, position) pairs per lane-group per round
    constexpr int RC = DEPTH;                              // value rows per thread requested up front
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, heads = a.heads, hd = H / heads;
    const int P = SELF ? a.t + 1 : a.S;
    const int C = H / VEC;
    const int npg = 256 / C > 0 ? 256 / C : 1;
    float* q_s = smem;                 // [H]
    float* sc = q_s + H;               // [heads][P]
    float* part = sc + ((heads * P + 3) & ~3);  // [npg][H]
    int* anc_s = reinterpret_cast<int*>(part + npg * H);  // [P] (self only)

    int n = blockIdx.x;
    if constexpr (!SELF) {
        // workgroups b and b+8 share an XCD (and its L2): hand the rows of one image to one XCD so that
        // only the first of them fetches the image's keys/values from beyond L2
        const int R = a.rows_per_image, imgs = a.N / R;
        if (R > 1 && (imgs & 7) == 0) {
            const int xcd = n & 7, idx = n >> 3;
            n = ((idx / R) * 8 + xcd) * R + idx % R;
        }
    }
    const int tid = threadIdx.x;
    const float scale = sqrtf(1.0f / (float)hd);
    const float* qrow = a.q + (long long)n * a.ldq;
    for (int i = tid; i < H; i += 256) q_s[i] = qrow[i] * scale;
    if constexpr (SELF) {
        T* kc = (T*)a.kcache + ((long long)n * a.t_cap + a.t) * H;
        T* vc = (T*)a.vcache + ((long long)n * a.t_cap + a.t) * H;
        for (int i = tid; i < H; i += 256) {
            kc[i] = from_f32<T>(qrow[H + i]);
            vc[i] = from_f32<T>(qrow[2 * H + i]);
        }
        for (int p = tid; p < a.t; p += 256) anc_s[p] = a.anc ? a.anc[(long long)n * a.t_cap + p] : n;
    }
    __syncthreads();

    // 16 bytes of the key / value row of position p at element offset off; position t of the self
    // form comes from the fresh projection rounded through T (what was just written to the cache)
    auto load_raw = [&](int p, int off, bool value) -> Vec16<T> {
        if constexpr (SELF) {
            if (p == a.t) {
                const float* src = qrow + (value ? 2 * H : H) + off;
                Vec16<T> r;
#pragma unroll
                for (int e = 0; e < VEC; ++e) r.v[e] = from_f32<T>(src[e]);
                return r;
            }
            return ld16((const T*)(value ? a.vcache : a.kcache) + ((long long)anc_s[p] * a.t_cap + p) * H + off);
        } else {
            return ld16((const T*)a.kv + ((long long)(n / a.rows_per_image) * a.S + p) * 2 * H + (value ? H : 0) + off);
        }
    };

    // value rows of the first phase-C round: requested now, consumed after the softmax
    const int pg = tid / C, cc = tid - pg * C;
    const bool c_active = pg < npg;
    Vec16<T> vpre[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int p = pg + i * npg;
        if (c_active && p < P) vpre[i] = load_raw(p, cc * VEC, true);
    }

    // phase A
    {
        const int grp = tid >> 4, gl = tid & 15;
        const int npairs = heads * P;
        const int chunks = hd / VEC;
        for (int pair0 = grp; pair0 < npairs; pair0 += 16 * RA) {
            Vec16<T> kraw[RA][CPL];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int pair = pair0 + 16 * i;
                if (pair < npairs) {
                    const int h = pair / P, p = pair - h * P;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = gl + 16 * j;
                        if (c < chunks) kraw[i][j] = load_raw(p, h * hd + c * VEC, false);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int pair = pair0 + 16 * i;
                if (pair < npairs) {  // uniform within the 16-lane group
                    const int h = pair / P;
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = gl + 16 * j;
                        if (c < chunks) {
                            const float* qq = q_s + h * hd + c * VEC;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) s = fmaf(qq[e], to_f32<T>(kraw[i][j].v[e]), s);
                        }
                    }
                    s += __shfl_xor(s, 8, 16);
                    s += __shfl_xor(s, 4, 16);
                    s += __shfl_xor(s, 2, 16);
                    s += __shfl_xor(s, 1, 16);
                    if (gl == 0) sc[pair] = s;
                }
            }
        }
    }
    __syncthreads();
    // phase B
    {
        const int wv = tid >> 6, lane = tid & 63;
        for (int h = wv; h < heads; h += 4) {
            float* row = sc + h * P;
            float m = -INFINITY;
            for (int p = lane; p < P; p += 64) m = fmaxf(m, row[p]);
            m = wave_max(m);
            float s = 0.f;
            for (int p = lane; p < P; p += 64) {
                const float e = expf(row[p] - m);
                row[p] = e;
                s += e;
            }
            s = wave_sum(s);
            const float inv = 1.0f / s;
            for (int p = lane; p < P; p += 64) row[p] *= inv;
        }
    }
    __syncthreads();
    if constexpr (!SELF) {
        if (a.attn_out) {
            const float hs = a.attn_scale / (float)heads;
            for (int p = tid; p < P; p += 256) {
                float s = 0.f;
                for (int h = 0; h < heads; ++h) s += sc[h * P + p];
                float* dst = a.attn_out + (long long)n * a.ld_attn + p;
                *dst = (a.accumulate ? *dst : 0.f) + s * hs;
            }
        }
    }
    // phase C
    if (c_active) {
        const int h = (cc * VEC) / hd;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int p = pg + i * npg;
            if (p < P) {
                const float wgt = sc[h * P + p];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(vpre[i].v[e]), acc[e]);
            }
        }
        for (int p = pg + RC * npg; p < P; p += npg) {  // positions beyond the prefetched round
            const Vec16<T> v = load_raw(p, cc * VEC, true);
            const float wgt = sc[h * P + p];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(v.v[e]), acc[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[pg * H + cc * VEC + e] = acc[e];
    }
    __syncthreads();
    T* orow = (T*)a.o + (long long)n * a.ld_o;
    for (int i = tid; i < H; i += 256) {
        float s = 0.f;
        for (int g = 0; g < npg; ++g) s += part[g * H + i];
        orow[i] = from_f32<T>(s);
    }
}

// Self-attention step for H == 64 chunks of 16 bytes and at most 32 cached positions: wave w owns
// positions w, w+4, ...; a key / value row is one fully coalesced 1 KiB wave-instruction, and every row
// of the prefix is requested up front (ancestry index, then the row: two dependent round trips for
// the whole launch, no barrier in front of them).
template <typename T>
__global__ __launch_bounds__(256) void self_attn_wide_kernel(const AttnArgs a) {
    constexpr int VEC = Vec16<T>::N, NI = 8, SP = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, heads = a.heads, t = a.t, P = t + 1;
    const int lph = 64 / heads;
    float* sc = smem;                 // [heads][SP]
    float* part = smem + heads * SP;  // [4][H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane / lph;
    // Workgroup b runs on XCD b % 8; the projection that wrote rows 16*rb .. 16*rb+15 ran on XCD rb % 8 and the
    // one that reads this kernel's output will too, so row n is handled on XCD (n / 16) % 8 when the row count
    // divides into 8 XCDs x 16 rows (a bijection of [0, N) then).
    int n = blockIdx.x;
    if ((gridDim.x & 127) == 0) {
        const int x = n & 7, j = n >> 3;
        n = (((j >> 4) << 3) + x) * 16 + (j & 15);
    }
    const float scale = sqrtf(1.0f / (float)(H / heads));
    const float* qrow = a.q + (long long)n * a.ldq + lane * VEC;
    Vec16<T> knew, vnew;  // position t: the fresh projection rounded through T, as the cache will hold it
    float q[VEC];
#pragma unroll
    for (int e4 = 0; e4 < VEC; e4 += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qrow + e4);
        const float4 kv = *reinterpret_cast<const float4*>(qrow + H + e4);
        const float4 vv = *reinterpret_cast<const float4*>(qrow + 2 * H + e4);
        q[e4] = qv.x * scale; q[e4 + 1] = qv.y * scale; q[e4 + 2] = qv.z * scale; q[e4 + 3] = qv.w * scale;
        knew.v[e4] = from_f32<T>(kv.x); knew.v[e4 + 1] = from_f32<T>(kv.y); knew.v[e4 + 2] = from_f32<T>(kv.z); knew.v[e4 + 3] = from_f32<T>(kv.w);
        vnew.v[e4] = from_f32<T>(vv.x); vnew.v[e4 + 1] = from_f32<T>(vv.y); vnew.v[e4 + 2] = from_f32<T>(vv.z); vnew.v[e4 + 3] = from_f32<T>(vv.w);
    }
    Vec16<T> kreg[NI], vreg[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + 4 * i;
        if (p < t) {
            const long long src = a.anc ? a.anc[(long long)n * a.t_cap + p] : n;
            kreg[i] = ld16((const T*)a.kcache + (src * a.t_cap + p) * H + lane * VEC);
            vreg[i] = ld16((const T*)a.vcache + (src * a.t_cap + p) * H + lane * VEC);
        } else if (p == t) {
            kreg[i] = knew;
            vreg[i] = vnew;
            st16((T*)a.kcache + ((long long)n * a.t_cap + t) * H + lane * VEC, knew);
            st16((T*)a.vcache + ((long long)n * a.t_cap + t) * H + lane * VEC, vnew);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + 4 * i;
        if (p < P) {
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v = fmaf(q[e], to_f32<T>(kreg[i].v[e]), v);
            for (int o = lph >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((lane & (lph - 1)) == 0) sc[h * SP + p] = v;
        }
    }
    __syncthreads();
    for (int hh = wave; hh < heads; hh += 4) {
        float* row = sc + hh * SP;
        const float v = lane < P ? row[lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = lane < P ? expf(v - m) : 0.f;
        const float tot = wave_sum(e);
        if (lane < P) row[lane] = e / tot;
    }
    __syncthreads();
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + 4 * i;
        if (p < P) {
            const float wgt = sc[h * SP + p];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, to_f32<T>(vreg[i].v[e]), acc[e]);
        }
    }
#pragma unroll
    for (int e4 = 0; e4 < VEC; e4 += 4)
        *reinterpret_cast<float4*>(part + wave * H + lane * VEC + e4) = make_float4(acc[e4], acc[e4 + 1], acc[e4 + 2], acc[e4 + 3]);
    __syncthreads();
    T* orow = (T*)a.o + (long long)n * a.ld_o;
    for (int i = tid; i < H; i += 256) orow[i] = from_f32<T>((part[i] + part[H + i]) + (part[2 * H + i] + part[3 * H + i]));
}

// Cross-attention for all rows of one image in one workgroup: the beams of an image attend to the
// same S keys/values, so each key / value row is loaded once and used for every beam (the per-row
// form re-reads 2*S*H elements per beam from L2, which is what bounds it).
constexpr int MAXR = MIC_MAX_BEAM;

template <typename T, int CPL>
__global__ __launch_bounds__(256) void cross_attn_image_kernel(const AttnArgs a) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int RA = 8 / CPL, RC = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, heads = a.heads, hd = H / heads, S = a.S, R = a.rows_per_image;
    const int C = H / VEC;
    const int npg = 256 / C > 0 ? 256 / C : 1;
    float* q_s = smem;                                  // [R][H]
    float* sc = q_s + R * H;                            // [R][heads][S]
    float* part = sc + ((R * heads * S + 3) & ~3);      // [npg][R][H]
    const int b = blockIdx.x, tid = threadIdx.x, n0 = b * R;
    const float scale = sqrtf(1.0f / (float)hd);
    for (int i = tid; i < R * H; i += 256) {
        const int j = i / H;
        q_s[i] = a.q[(long long)(n0 + j) * a.ldq + (i - j * H)] * scale;
    }
    const T* kvb = (const T*)a.kv + (long long)b * S * 2 * H;
    const int pg = tid / C, cc = tid - pg * C;
    const bool c_active = pg < npg;
    Vec16<T> vpre[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int p = pg + i * npg;
        if (c_active && p < S) vpre[i] = ld16(kvb + (long long)p * 2 * H + H + cc * VEC);
    }
    __syncthreads();
    {   // scores: one key chunk load serves the R queries
        const int grp = tid >> 4, gl = tid & 15;
        const int npairs = heads * S, chunks = hd / VEC;
        for (int pair0 = grp; pair0 < npairs; pair0 += 16 * RA) {
            Vec16<T> kraw[RA][CPL];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int pair = pair0 + 16 * i;
                if (pair < npairs) {
                    const int h = pair / S, p = pair - h * S;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = gl + 16 * j;
                        if (c < chunks) kraw[i][j] = ld16(kvb + (long long)p * 2 * H + h * hd + c * VEC);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int pair = pair0 + 16 * i;
                if (pair < npairs) {
                    const int h = pair / S, p = pair - h * S;
                    float sum[MAXR];
#pragma unroll
                    for (int rr = 0; rr < MAXR; ++rr) sum[rr] = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = gl + 16 * j;
                        if (c < chunks) {
                            float kf[VEC];
#pragma unroll
                            for (int e = 0; e < VEC; ++e) kf[e] = to_f32<T>(kraw[i][j].v[e]);
#pragma unroll
                            for (int rr = 0; rr < MAXR; ++rr) {
                                if (rr < R) {
                                    const float* qq = q_s + rr * H + h * hd + c * VEC;
#pragma unroll
                                    for (int e = 0; e < VEC; ++e) sum[rr] = fmaf(qq[e], kf[e], sum[rr]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < MAXR; ++rr) {
                        if (rr < R) {
                            float v = sum[rr];
                            v += __shfl_xor(v, 8, 16);
                            v += __shfl_xor(v, 4, 16);
                            v += __shfl_xor(v, 2, 16);
                            v += __shfl_xor(v, 1, 16);
                            if (gl == 0) sc[(rr * heads + h) * S + p] = v;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    {   // softmax of each (row, head)
        const int wv = tid >> 6, lane = tid & 63;
        for (int rh = wv; rh < R * heads; rh += 4) {
            float* row = sc + rh * S;
            float m = -INFINITY;
            for (int p = lane; p < S; p += 64) m = fmaxf(m, row[p]);
            m = wave_max(m);
            float sum = 0.f;
            for (int p = lane; p < S; p += 64) {
                const float e = expf(row[p] - m);
                row[p] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int p = lane; p < S; p += 64) row[p] *= inv;
        }
    }
    __syncthreads();
    if (a.attn_out) {
        const float hs = a.attn_scale / (float)heads;
        for (int i = tid; i < R * S; i += 256) {
            const int rr = i / S, p = i - rr * S;
            float sum = 0.f;
            for (int h = 0; h < heads; ++h) sum += sc[(rr * heads + h) * S + p];
            float* dst = a.attn_out + (long long)(n0 + rr) * a.ld_attn + p;
            *dst = (a.accumulate ? *dst : 0.f) + sum * hs;
        }
    }
    if (c_active) {
        const int h = (cc * VEC) / hd;
        float acc[MAXR][VEC];
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[rr][e] = 0.f;
        auto use = [&](int p, const Vec16<T>& v) {
            float vf[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) vf[e] = to_f32<T>(v.v[e]);
#pragma unroll
            for (int rr = 0; rr < MAXR; ++rr) {
                if (rr < R) {
                    const float wgt = sc[(rr * heads + h) * S + p];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[rr][e] = fmaf(wgt, vf[e], acc[rr][e]);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int p = pg + i * npg;
            if (p < S) use(p, vpre[i]);
        }
        for (int p = pg + RC * npg; p < S; p += npg) use(p, ld16(kvb + (long long)p * 2 * H + H + cc * VEC));
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr) {
            if (rr < R) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) part[(pg * R + rr) * H + cc * VEC + e] = acc[rr][e];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < R * H; i += 256) {
        const int rr = i / H, col = i - rr * H;
        float sum = 0.f;
        for (int g = 0; g < npg; ++g) sum += part[(g * R + rr) * H + col];
        ((T*)a.o)[(long long)(n0 + rr) * a.ld_o + col] = from_f32<T>(sum);
    }
}

// Cross-attention for all beams of one image by a workgroup of 16 waves, for H == 64 chunks of 16
// bytes (one wave-instruction = one whole key or value row, 1 KiB, fully coalesced) and S <= 64.
// Every key and value row of the image is requested once, up front, by the wave that owns the
// position (p = wave + 16 i), so the launch pays ONE memory round trip; scores, softmax and the
// weighted sum then run on registers and LDS, and each row is used for every beam.  Beams are taken
// in groups of RG to bound registers.
template <typename T, int RG>
__global__ __launch_bounds__(1024) void cross_attn_wide_kernel(const AttnArgs a) {
    constexpr int VEC = Vec16<T>::N, NI = 4, SP = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, heads = a.heads, S = a.S, R = a.rows_per_image;
    const int lph = 64 / heads;  // lanes per head
    float* sc = smem;                       // [RG][heads][SP]
    float* part = smem + RG * heads * SP;   // [8][RG][H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, n0 = b * R, h = lane / lph;
    const float scale = sqrtf(1.0f / (float)(H / heads));
    const T* kvb = (const T*)a.kv + (long long)b * S * 2 * H + lane * VEC;
    Vec16<T> kreg[NI], vreg[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = wave + 16 * i;
        if (p < S) {
            kreg[i] = ld16(kvb + (long long)p * 2 * H);
            vreg[i] = ld16(kvb + (long long)p * 2 * H + H);
        }
    }
    for (int r0 = 0; r0 < R; r0 += RG) {
        const int rn = R - r0 < RG ? R - r0 : RG;
        float q[RG][VEC];
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            const float* qp = a.q + (long long)(n0 + r0 + (rr < rn ? rr : 0)) * a.ldq + lane * VEC;
#pragma unroll
            for (int e4 = 0; e4 < VEC; e4 += 4) {
                const float4 v = *reinterpret_cast<const float4*>(qp + e4);
                q[rr][e4] = v.x * scale; q[rr][e4 + 1] = v.y * scale; q[rr][e4 + 2] = v.z * scale; q[rr][e4 + 3] = v.w * scale;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
                float kf[VEC], sum[RG];
#pragma unroll
                for (int e = 0; e < VEC; ++e) kf[e] = to_f32<T>(kreg[i].v[e]);
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    float v = 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v = fmaf(q[rr][e], kf[e], v);
                    for (int o = lph >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    sum[rr] = v;
                }
                if ((lane & (lph - 1)) == 0) {
#pragma unroll
                    for (int rr = 0; rr < RG; ++rr)
                        if (rr < rn) sc[(rr * heads + h) * SP + p] = sum[rr];
                }
            }
        }
        __syncthreads();
        for (int rh = wave; rh < rn * heads; rh += 16) {  // softmax of one (beam, head) per wave, lane = position
            float* row = sc + rh * SP;
            const float v = lane < S ? row[lane] : -INFINITY;
            const float m = wave_max(v);
            const float e = lane < S ? expf(v - m) : 0.f;
            const float tot = wave_sum(e);
            if (lane < S) row[lane] = e / tot;
        }
        __syncthreads();
        if (a.attn_out) {
            const float hs = a.attn_scale / (float)heads;
            for (int i = tid; i < rn * S; i += 1024) {
                const int rr = i / S, p = i - rr * S;
                float sum = 0.f;
                for (int hh = 0; hh < heads; ++hh) sum += sc[(rr * heads + hh) * SP + p];
                float* dst = a.attn_out + (long long)(n0 + r0 + rr) * a.ld_attn + p;
                *dst = (a.accumulate ? *dst : 0.f) + sum * hs;
            }
        }
        float acc[RG][VEC];
#pragma unroll
        for (int rr = 0; rr < RG; ++rr)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[rr][e] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
                float vf[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) vf[e] = to_f32<T>(vreg[i].v[e]);
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    const float wgt = rr < rn ? sc[(rr * heads + h) * SP + p] : 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[rr][e] = fmaf(wgt, vf[e], acc[rr][e]);
                }
            }
        }
        // 16 partial sums per output: waves 8..15 hand theirs to waves 0..7, which publish the pair sums
        float* mine = part + ((wave & 7) * RG) * H + lane * VEC;
        if (wave >= 8) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4)
                    *reinterpret_cast<float4*>(mine + rr * H + e4) = make_float4(acc[rr][e4], acc[rr][e4 + 1], acc[rr][e4 + 2], acc[rr][e4 + 3]);
        }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4) {
                    const float4 o = *reinterpret_cast<const float4*>(mine + rr * H + e4);
                    *reinterpret_cast<float4*>(mine + rr * H + e4) =
                        make_float4(acc[rr][e4] + o.x, acc[rr][e4 + 1] + o.y, acc[rr][e4 + 2] + o.z, acc[rr][e4 + 3] + o.w);
                }
        }
        __syncthreads();
        for (int i = tid; i < rn * H; i += 1024) {
            const int rr = i / H, col = i - rr * H;
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sum += part[(g * RG + rr) * H + col];
            ((T*)a.o)[(long long)(n0 + r0 + rr) * a.ld_o + col] = from_f32<T>(sum);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Whole cross-attention sub-layer of a post-LN decoder layer in one launch, for few heads:
//     y = LN(r) + sum_h softmax_s( LN(r) . K'_h[s] + c_h[s] ) V'_h[s] + b_o          (+ row statistics of y)
// with the query projection folded into the per-image keys, K'_h[s] = W_q,h^T k_h[s] / sqrt(d) and
// c_h[s] = b_q,h . k_h[s] / sqrt(d), and the output projection folded into the values,
// V'_h[s] = W_o[:, h] v_h[s]; both are computed once per image next to the K/V projection.  The two
// projection GEMMs of the sub-layer and their launches disappear; the price is H instead of H/heads
// elements per (position, head) of K' and V'.  Same decomposition as cross_attn_wide_kernel: 16 waves per
// image, wave w owns positions w, w+16, ..., every K' row requested up front, V' rows as soon as the
// scores are out (in flight during the softmax).
struct FoldArgs {
    const float* raw;      // [N][ld] pre-LayerNorm stream r (y may alias it)
    const float* stats;    // [N][parts][2]
    const float *gamma, *beta;
    int ld, parts;
    const void *kf, *vf;   // [B][S][heads][H]
    const float* c;        // [B][S][heads]
    const float* bo;       // [H]
    float* y;              // [N][H]
    float* stats_out;      // [N][parts][2]
    float* attn_out;
    int ld_attn;
    float attn_scale;
    int accumulate, S, R, N, H;
};

// MULTI: more beams than one register group; the K' rows are then re-requested per group (they come from L2
// after the first) so that K' and V' never have to be held in registers together.
template <typename T, int RG, int HEADS, bool MULTI>
__global__ __launch_bounds__(1024) void cross_fold_kernel(const FoldArgs a) {
    constexpr int VEC = Vec16<T>::N, NI = 4, SP = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, S = a.S, R = a.R;
    float* xs = smem;                           // [R][H] LayerNorm(r)
    float* sc = xs + R * H;                     // [RG][HEADS][SP]
    float* part = sc + RG * HEADS * SP;         // [8][RG][H]
    float* rstat = part + 8 * RG * H;           // [R][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, n0 = b * R;
    const T* kfb = (const T*)a.kf + (long long)b * S * HEADS * H + lane * VEC;
    const T* vfb = (const T*)a.vf + (long long)b * S * HEADS * H + lane * VEC;
    Vec16<T> kreg[NI][HEADS];
    auto load_keys = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
#pragma unroll
                for (int h = 0; h < HEADS; ++h) kreg[i][h] = ld16(kfb + ((long long)p * HEADS + h) * H);
            }
        }
    };
    if (!MULTI) load_keys();  // in flight during the LayerNorm prologue
    if (tid < R) {  // mean and 1/sqrt(var + eps) of row tid from its per-part (mean, M2) pairs
        const float* st = a.stats + (long long)(n0 + tid) * a.parts * 2;
        float n = 0.f, mu = 0.f, m2 = 0.f;
        for (int p = 0; p < a.parts; ++p) {
            const float nb = (float)min(128, H - 128 * p), mb = st[2 * p], sb = st[2 * p + 1];
            const float nt = n + nb, d = mb - mu;
            mu += d * (nb / nt);
            m2 += sb + d * d * (n * nb / nt);
            n = nt;
        }
        rstat[2 * tid] = mu;
        rstat[2 * tid + 1] = 1.0f / sqrtf(m2 / n + 1e-5f);
    }
    __syncthreads();
    for (int i = tid; i < R * (H / 4); i += 1024) {
        const int r = i / (H / 4), c4 = (i - r * (H / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(a.raw + (long long)(n0 + r) * a.ld + c4);
        const float4 g = *reinterpret_cast<const float4*>(a.gamma + c4), be = *reinterpret_cast<const float4*>(a.beta + c4);
        const float mean = rstat[2 * r], rstd = rstat[2 * r + 1];
        *reinterpret_cast<float4*>(xs + r * H + c4) = make_float4((v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y,
                                                                  (v.z - mean) * rstd * g.z + be.z, (v.w - mean) * rstd * g.w + be.w);
    }
    __syncthreads();
    for (int r0 = 0; r0 < (MULTI ? R : 1); r0 += RG) {
        const int rn = R - r0 < RG ? R - r0 : RG;
        if (MULTI) load_keys();
        float q[RG][VEC];
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            const float* qp = xs + (r0 + (rr < rn ? rr : 0)) * H + lane * VEC;
#pragma unroll
            for (int e4 = 0; e4 < VEC; e4 += 4) {
                const float4 v = *reinterpret_cast<const float4*>(qp + e4);
                q[rr][e4] = v.x; q[rr][e4 + 1] = v.y; q[rr][e4 + 2] = v.z; q[rr][e4 + 3] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
#pragma unroll
                for (int h = 0; h < HEADS; ++h) {
                    float kf[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) kf[e] = to_f32<T>(kreg[i][h].v[e]);
                    const float cc = a.c[((long long)b * S + p) * HEADS + h];
#pragma unroll
                    for (int rr = 0; rr < RG; ++rr) {
                        float v = 0.f;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) v = fmaf(q[rr][e], kf[e], v);
                        v = wave_sum(v);
                        if (lane == 0 && rr < rn) sc[(rr * HEADS + h) * SP + p] = v + cc;
                    }
                }
            }
        }
        // value rows: requested now, consumed after the softmax
        Vec16<T> vreg[NI][HEADS];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
#pragma unroll
                for (int h = 0; h < HEADS; ++h) vreg[i][h] = ld16(vfb + ((long long)p * HEADS + h) * H);
            }
        }
        __syncthreads();
        for (int rh = wave; rh < rn * HEADS; rh += 16) {
            float* row = sc + rh * SP;
            const float v = lane < S ? row[lane] : -INFINITY;
            const float m = wave_max(v);
            const float e = lane < S ? expf(v - m) : 0.f;
            const float tot = wave_sum(e);
            if (lane < S) row[lane] = e / tot;
        }
        __syncthreads();
        if (a.attn_out) {
            const float hs = a.attn_scale / (float)HEADS;
            for (int i = tid; i < rn * S; i += 1024) {
                const int rr = i / S, p = i - rr * S;
                float sum = 0.f;
#pragma unroll
                for (int hh = 0; hh < HEADS; ++hh) sum += sc[(rr * HEADS + hh) * SP + p];
                float* dst = a.attn_out + (long long)(n0 + r0 + rr) * a.ld_attn + p;
                *dst = (a.accumulate ? *dst : 0.f) + sum * hs;
            }
        }
        float acc[RG][VEC];
#pragma unroll
        for (int rr = 0; rr < RG; ++rr)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[rr][e] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = wave + 16 * i;
            if (p < S) {
#pragma unroll
                for (int h = 0; h < HEADS; ++h) {
                    float vf[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) vf[e] = to_f32<T>(vreg[i][h].v[e]);
#pragma unroll
                    for (int rr = 0; rr < RG; ++rr) {
                        const float wgt = rr < rn ? sc[(rr * HEADS + h) * SP + p] : 0.f;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[rr][e] = fmaf(wgt, vf[e], acc[rr][e]);
                    }
                }
            }
        }
        float* mine = part + ((wave & 7) * RG) * H + lane * VEC;
        if (wave >= 8) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4)
                    *reinterpret_cast<float4*>(mine + rr * H + e4) = make_float4(acc[rr][e4], acc[rr][e4 + 1], acc[rr][e4 + 2], acc[rr][e4 + 3]);
        }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4) {
                    const float4 o = *reinterpret_cast<const float4*>(mine + rr * H + e4);
                    *reinterpret_cast<float4*>(mine + rr * H + e4) =
                        make_float4(acc[rr][e4] + o.x, acc[rr][e4 + 1] + o.y, acc[rr][e4 + 2] + o.z, acc[rr][e4 + 3] + o.w);
                }
        }
        __syncthreads();
        // y = attention + b_o + LN(r), and its statistics per 128 columns: one wave per (row, part), 2 columns per lane
        for (int pair = wave; pair < rn * a.parts; pair += 16) {
            const int rr = pair / a.parts, pp = pair - rr * a.parts;
            const int col = pp * 128 + lane * 2;
            const bool ok = col < H;  // H is a multiple of 4, so both columns or none
            float v0 = 0.f, v1 = 0.f;
            if (ok) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float2 t = *reinterpret_cast<const float2*>(part + (g * RG + rr) * H + col);
                    v0 += t.x; v1 += t.y;
                }
                v0 += a.bo[col] + xs[(r0 + rr) * H + col];
                v1 += a.bo[col + 1] + xs[(r0 + rr) * H + col + 1];
                *reinterpret_cast<float2*>(a.y + (long long)(n0 + r0 + rr) * H + col) = make_float2(v0, v1);
            }
            const float cnt = (float)min(128, H - 128 * pp);
            const float mean = wave_sum(ok ? v0 + v1 : 0.f) / cnt;
            const float d0 = ok ? v0 - mean : 0.f, d1 = ok ? v1 - mean : 0.f;
            const float m2 = wave_sum(d0 * d0 + d1 * d1);
            if (lane == 0 && a.stats_out) {
                float* dst = a.stats_out + ((long long)(n0 + r0 + rr) * a.parts + pp) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
        }
        __syncthreads();
    }
}

// c[row][h] = qb[h*hd ..] . k[row][h*hd ..]: one wave per (key row, head)
template <typename T>
__global__ __launch_bounds__(256) void cross_fold_bias_kernel(const T* __restrict__ kv, const float* __restrict__ qb, float* __restrict__ c,
                                                              int rows, int H, int heads) {
    constexpr int VEC = Vec16<T>::N;
    const int id = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (id >= rows * heads) return;
    const int row = id / heads, h = id - row * heads, hd = H / heads;
    float s = 0.f;
    for (int ch = lane; ch < hd / VEC; ch += 64) {
        const Vec16<T> k = ld16(kv + (long long)row * 2 * H + h * hd + ch * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s = fmaf(qb[h * hd + ch * VEC + e], to_f32<T>(k.v[e]), s);
    }
    s = wave_sum(s);
    if (lane == 0) c[id] = s;
}

// c[row][h] = feats[row] . u[h] + c0[h]: the same per-position constant computed from the image tokens themselves
// (u_h = W_k,h^T b_q,h / sqrt d), for callers that project the folded keys straight from the tokens and never form k.
template <typename T>
__global__ __launch_bounds__(256) void cross_fold_c_kernel(const T* __restrict__ feats, const float* __restrict__ u, const float* __restrict__ c0,
                                                           float* __restrict__ c, int rows, int H, int heads) {
    constexpr int VEC = Vec16<T>::N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s[MIC_FOLD_MAX_HEADS];
#pragma unroll
    for (int h = 0; h < MIC_FOLD_MAX_HEADS; ++h) s[h] = 0.f;
    for (int ch = lane; ch < H / VEC; ch += 64) {
        const Vec16<T> f = ld16(feats + (long long)row * H + ch * VEC);
#pragma unroll
        for (int h = 0; h < MIC_FOLD_MAX_HEADS; ++h)
            if (h < heads) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[h] = fmaf(u[h * H + ch * VEC + e], to_f32<T>(f.v[e]), s[h]);
            }
    }
#pragma unroll
    for (int h = 0; h < MIC_FOLD_MAX_HEADS; ++h)
        if (h < heads) {
            const float t = wave_sum(s[h]);
            if (lane == 0) c[(long long)row * heads + h] = t + c0[h];
        }
}

template <typename T, int RG, int HEADS, bool MULTI> int launch_cross_fold_rg(const FoldArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)a.R * a.H + RG * HEADS * 64 + 8 * RG * (size_t)a.H + 2 * a.R);
    MIC_ARG_CHECK(lds <= 150 * 1024, "mic_cross_attn_folded: %zu bytes of LDS needed", lds);
    static std::atomic<uint64_t> granted{0};  // one bit per device; lanes call from several host threads
    MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&cross_fold_kernel<T, RG, HEADS, MULTI>), 150 * 1024, granted));
    MIC_LAUNCH((cross_fold_kernel<T, RG, HEADS, MULTI>), dim3(a.N / a.R), dim3(1024), lds, s, a);
    return 0;
}
template <typename T, int HEADS> int launch_cross_fold(const FoldArgs& a, hipStream_t s) {
    if (a.R <= 2) return launch_cross_fold_rg<T, 2, HEADS, false>(a, s);
    if (a.R == 3) return launch_cross_fold_rg<T, 3, HEADS, false>(a, s);
    if (a.R == 4) return launch_cross_fold_rg<T, 4, HEADS, false>(a, s);
    if (a.R == 8 || a.R == 7) return launch_cross_fold_rg<T, 4, HEADS, true>(a, s);
    return launch_cross_fold_rg<T, 3, HEADS, true>(a, s);
}

template <typename T, int RG> int launch_cross_wide_rg(const AttnArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * (RG * (size_t)a.heads * 64 + 8 * RG * (size_t)a.H);
    static std::atomic<uint64_t> granted{0};  // one bit per device; lanes call from several host threads
    MIC_TRY(mic_grant_dynamic_lds(reinterpret_cast<const void*>(&cross_attn_wide_kernel<T, RG>), 150 * 1024, granted));
    MIC_LAUNCH((cross_attn_wide_kernel<T, RG>), dim3(a.N / a.rows_per_image), dim3(1024), lds, s, a);
    return 0;
}
template <typename T> int launch_cross_wide(const AttnArgs& a, hipStream_t s) {
    const int R = a.rows_per_image;  // group size: few idle slots in the last group, at most 4 beams in registers
    if (R == 2) return launch_cross_wide_rg<T, 2>(a, s);
    if (R % 4 == 0 || R == 7) return launch_cross_wide_rg<T, 4>(a, s);
    return launch_cross_wide_rg<T, 3>(a, s);
}

template <typename T> int launch_cross_image(const AttnArgs& a, hipStream_t s, size_t lds) {
    constexpr int VEC = Vec16<T>::N;
    const int cpl = (a.H / a.heads / VEC + 15) / 16;
    const dim3 grid(a.N / a.rows_per_image);
    if (cpl <= 1) MIC_LAUNCH((cross_attn_image_kernel<T, 1>), grid, dim3(256), lds, s, a);
    else if (cpl <= 2) MIC_LAUNCH((cross_attn_image_kernel<T, 2>), grid, dim3(256), lds, s, a);
    else if (cpl <= 4) MIC_LAUNCH((cross_attn_image_kernel<T, 4>), grid, dim3(256), lds, s, a);
    else MIC_LAUNCH((cross_attn_image_kernel<T, 8>), grid, dim3(256), lds, s, a);
    return 0;
}

template <typename T, bool SELF> int launch_attn(const AttnArgs& a, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const int P = SELF ? a.t + 1 : a.S;
    const int C = a.H / VEC;
    const int npg = 256 / C > 0 ? 256 / C : 1;
    const size_t lds = sizeof(float) * ((size_t)a.H + ((a.heads * P + 3) & ~3) + (size_t)npg * a.H) + sizeof(int) * (SELF ? P : 0);
    MIC_ARG_CHECK(lds <= 64 * 1024, "attention step: %zu bytes of LDS needed (H=%d heads=%d positions=%d)", lds, a.H, a.heads, P);
    const int cpl = (a.H / a.heads / VEC + 15) / 16;
    if (cpl <= 1) MIC_LAUNCH((attn_step_kernel<T, SELF, 1>), dim3(a.N), dim3(256), lds, s, a);
    else if (cpl <= 2) MIC_LAUNCH((attn_step_kernel<T, SELF, 2>), dim3(a.N), dim3(256), lds, s, a);
    else if (cpl <= 4) MIC_LAUNCH((attn_step_kernel<T, SELF, 4>), dim3(a.N), dim3(256), lds, s, a);
    else MIC_LAUNCH((attn_step_kernel<T, SELF, 8>), dim3(a.N), dim3(256), lds, s, a);
    return 0;
}

int check_attn_dims(const char* who, int dtype, int N, int H, int heads) {
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "%s: bad dtype %d", who, dtype);
    MIC_ARG_CHECK(N > 0 && H > 0 && heads > 0 && H % heads == 0, "%s: bad sizes N=%d H=%d heads=%d", who, N, H, heads);
    const int vec = 16 / (int)mic_dtype_size(dtype);
    MIC_ARG_CHECK((H / heads) % vec == 0, "%s: head_dim=%d must be a multiple of %d", who, H / heads, vec);
    MIC_ARG_CHECK(H / vec <= 256, "%s: H=%d too large for one workgroup per row", who, H);
    MIC_ARG_CHECK(H / heads / vec <= 128, "%s: head_dim=%d too large", who, H / heads);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__global__ void lstm_cell_kernel(const float* __restrict__ gates, float* __restrict__ c, float* __restrict__ h, T* ht0, int ld0,
                                 T* ht1, int ld1, int N, int Hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, j = i - n * Hd;
    const float* g = gates + (long long)n * 4 * Hd;
    const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[Hd + j]), gg = tanhf(g[2 * Hd + j]), og = sigmoidf_(g[3 * Hd + j]);
    const float cn = __builtin_fmaf(fg, c[i], ig * gg);  // spelt out: the same rounding wherever this cell is written (gemm_ln.hip has it too)
    const float hn = og * tanhf(cn);
    c[i] = cn;
    h[i] = hn;
    if (ht0) ht0[(long long)n * ld0 + j] = from_f32<T>(hn);
    if (ht1) ht1[(long long)n * ld1 + j] = from_f32<T>(hn);
}

// The cell of the top layer with the LayerNorm that follows it (models/intermediate.py:63,65): one wave per row, 4 rows
// per block, the row's h' kept in registers between the two.  Column order and arithmetic of the normalisation are those
// of layernorm_kernel.  NV = float4 per lane (Hd <= 256 * NV).
template <typename T, int NV>
__global__ void lstm_cell_ln_kernel(const float* __restrict__ gates, float* __restrict__ c, float* __restrict__ h, T* ht0, int ld0,
                                    const float* __restrict__ lg, const float* __restrict__ lb, float* hn, T* hnt, int ld_hnt, int N,
                                    int Hd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* g = gates + (long long)row * 4 * Hd;
    float4 hv[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane * 4 + 256 * k;
        hv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Hd) {
            const float4 gi = *reinterpret_cast<const float4*>(g + j), gf = *reinterpret_cast<const float4*>(g + Hd + j);
            const float4 gc = *reinterpret_cast<const float4*>(g + 2 * Hd + j), go = *reinterpret_cast<const float4*>(g + 3 * Hd + j);
            float4 cv = *reinterpret_cast<const float4*>(c + (long long)row * Hd + j);
            cv.x = __builtin_fmaf(sigmoidf_(gf.x), cv.x, sigmoidf_(gi.x) * tanhf(gc.x));
            cv.y = __builtin_fmaf(sigmoidf_(gf.y), cv.y, sigmoidf_(gi.y) * tanhf(gc.y));
            cv.z = __builtin_fmaf(sigmoidf_(gf.z), cv.z, sigmoidf_(gi.z) * tanhf(gc.z));
            cv.w = __builtin_fmaf(sigmoidf_(gf.w), cv.w, sigmoidf_(gi.w) * tanhf(gc.w));
            hv[k] = make_float4(sigmoidf_(go.x) * tanhf(cv.x), sigmoidf_(go.y) * tanhf(cv.y), sigmoidf_(go.z) * tanhf(cv.z),
                                sigmoidf_(go.w) * tanhf(cv.w));
            *reinterpret_cast<float4*>(c + (long long)row * Hd + j) = cv;
            *reinterpret_cast<float4*>(h + (long long)row * Hd + j) = hv[k];
            if (ht0) {
                T* t = ht0 + (long long)row * ld0 + j;
                t[0] = from_f32<T>(hv[k].x); t[1] = from_f32<T>(hv[k].y); t[2] = from_f32<T>(hv[k].z); t[3] = from_f32<T>(hv[k].w);
            }
            s += ln_sum4(hv[k].x, hv[k].y, hv[k].z, hv[k].w);
        }
    }
    const float mean = wave_sum(s) / (float)Hd;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane * 4 + 256 * k < Hd) q += ln_sq4(hv[k].x, hv[k].y, hv[k].z, hv[k].w, mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)Hd + 1e-5f);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane * 4 + 256 * k;
        if (j >= Hd) continue;
        const float4 gg = *reinterpret_cast<const float4*>(lg + j), bb = *reinterpret_cast<const float4*>(lb + j);
        float4 o;
        o.x = ln_apply(hv[k].x, mean, rstd, gg.x, bb.x);
        o.y = ln_apply(hv[k].y, mean, rstd, gg.y, bb.y);
        o.z = ln_apply(hv[k].z, mean, rstd, gg.z, bb.z);
        o.w = ln_apply(hv[k].w, mean, rstd, gg.w, bb.w);
        if (hn) *reinterpret_cast<float4*>(hn + (long long)row * Hd + j) = o;
        if (hnt) {
            T* t = hnt + (long long)row * ld_hnt + j;
            t[0] = from_f32<T>(o.x); t[1] = from_f32<T>(o.y); t[2] = from_f32<T>(o.z); t[3] = from_f32<T>(o.w);
        }
    }
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int32_t* __restrict__ idx,
                                   int rows, int words) {
    const long long total = (long long)rows * words;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / words), w = (int)(i - (long long)r * words);
        dst[i] = src[(long long)idx[r] * words + w];
    }
}

struct GatherMulti {
    const uint32_t* src[MIC_GATHER_MAX];
    uint32_t* dst[MIC_GATHER_MAX];
    int words[MIC_GATHER_MAX];
};

// blockIdx.y names the buffer: one launch moves the survivors' rows of every buffer
__global__ void gather_rows_multi_kernel(const GatherMulti g, const int32_t* __restrict__ idx, int rows) {
    const uint32_t* __restrict__ src = g.src[blockIdx.y];
    uint32_t* __restrict__ dst = g.dst[blockIdx.y];
    const int words = g.words[blockIdx.y];
    const long long total = (long long)rows * words;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / words), w = (int)(i - (long long)r * words);
        dst[i] = src[(long long)idx[r] * words + w];
    }
}

}  // namespace

extern "C" int mic_embed_pos(const int32_t* ids, int ld_ids, int pos, const float* tok_emb, const float* pos_emb, float* x,
                             void* xt, int ld_xt, int dtype, int N, int H, int vocab, void* stream) {
    MIC_ARG_CHECK(ids && tok_emb && (x || xt), "mic_embed_pos: null pointer");
    MIC_ARG_CHECK(N > 0 && H > 0 && H % 4 == 0 && vocab > 0 && pos >= 0, "mic_embed_pos: bad sizes N=%d H=%d", N, H);
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_embed_pos: bad dtype %d", dtype);
    const int total = N * (H / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(embed_pos_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, s, ids, ld_ids, pos, tok_emb, pos_emb, x, (bf16_t*)xt, ld_xt, N, H, vocab);
    else
        MIC_LAUNCH(embed_pos_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, ids, ld_ids, pos, tok_emb, pos_emb, x, (float*)xt, ld_xt, N, H, vocab);
    return 0;
}

extern "C" int mic_layernorm(const float* x, const float* gamma, const float* beta, float* y, void* yt, int ld_yt, int dtype,
                             int N, int H, void* stream) {
    MIC_ARG_CHECK(x && gamma && beta && (y || yt), "mic_layernorm: null pointer");
    MIC_ARG_CHECK(N > 0 && H > 0 && H % 4 == 0, "mic_layernorm: bad sizes N=%d H=%d", N, H);
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_layernorm: bad dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(layernorm_kernel<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, s, x, gamma, beta, (const float*)nullptr, y, (bf16_t*)yt, ld_yt, N, H);
    else
        MIC_LAUNCH(layernorm_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, x, gamma, beta, (const float*)nullptr, y, (float*)yt, ld_yt, N, H);
    return 0;
}

extern "C" int mic_layernorm_add(const float* u, const float* gamma, const float* beta, const float* res, float* y, void* yt,
                                 int ld_yt, int dtype, int N, int H, void* stream) {
    MIC_ARG_CHECK(u && gamma && beta && res && (y || yt), "mic_layernorm_add: null pointer");
    MIC_ARG_CHECK(N > 0 && H > 0 && H % 4 == 0, "mic_layernorm_add: bad sizes N=%d H=%d", N, H);
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_layernorm_add: bad dtype %d", dtype);
    MIC_ARG_CHECK(u != y, "mic_layernorm_add: the branch input must not alias the output");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(layernorm_kernel<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, s, u, gamma, beta, res, y, (bf16_t*)yt, ld_yt, N, H);
    else
        MIC_LAUNCH(layernorm_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, u, gamma, beta, res, y, (float*)yt, ld_yt, N, H);
    return 0;
}

extern "C" int mic_self_attn_step(const float* qkv, void* kcache, void* vcache, const int32_t* anc, int t, int t_cap, void* o,
                                  int ld_o, int dtype, int N, int H, int heads, void* stream) {
    MIC_ARG_CHECK(qkv && kcache && vcache && o, "mic_self_attn_step: null pointer");
    MIC_TRY(check_attn_dims("mic_self_attn_step", dtype, N, H, heads));
    MIC_ARG_CHECK(t >= 0 && t < t_cap && ld_o >= H, "mic_self_attn_step: t=%d t_cap=%d ld_o=%d", t, t_cap, ld_o);
    AttnArgs a{};
    a.q = qkv; a.ldq = 3 * H; a.kcache = kcache; a.vcache = vcache; a.anc = anc; a.t = t; a.t_cap = t_cap;
    a.o = o; a.ld_o = ld_o; a.N = N; a.H = H; a.heads = heads;
    hipStream_t s = (hipStream_t)stream;
    {
        static const bool no_wide = getenv("MIC_SELF_NO_WIDE") != nullptr;
        const int vec = 16 / (int)mic_dtype_size(dtype);
        if (!no_wide && H == 64 * vec && t_cap <= 32 && heads <= 64 && (heads & (heads - 1)) == 0 && ((uintptr_t)qkv & 15) == 0 &&
            ((uintptr_t)kcache & 15) == 0 && ((uintptr_t)vcache & 15) == 0) {
            const size_t lds = sizeof(float) * ((size_t)heads * 32 + 4 * (size_t)H);
            if (dtype == MIC_BF16) MIC_LAUNCH(self_attn_wide_kernel<bf16_t>, dim3(N), dim3(256), lds, s, a);
            else MIC_LAUNCH(self_attn_wide_kernel<float>, dim3(N), dim3(256), lds, s, a);
            return 0;
        }
    }
    return dtype == MIC_BF16 ? launch_attn<bf16_t, true>(a, s) : launch_attn<float, true>(a, s);
}

extern "C" int mic_cross_attn_step(const float* q, int ldq, const void* kv, int S, int rows_per_image, void* o, int ld_o,
                                   float* attn_out, int ld_attn, float attn_scale, int accumulate, int dtype, int N, int H,
                                   int heads, void* stream) {
    MIC_ARG_CHECK(q && kv && o, "mic_cross_attn_step: null pointer");
    MIC_TRY(check_attn_dims("mic_cross_attn_step", dtype, N, H, heads));
    MIC_ARG_CHECK(S > 0 && rows_per_image > 0 && ldq >= H && ld_o >= H, "mic_cross_attn_step: bad sizes S=%d", S);
    MIC_ARG_CHECK(!attn_out || ld_attn >= S, "mic_cross_attn_step: ld_attn=%d < S=%d", ld_attn, S);
    AttnArgs a{};
    a.q = q; a.ldq = ldq; a.kv = kv; a.S = S; a.rows_per_image = rows_per_image; a.o = o; a.ld_o = ld_o;
    a.attn_out = attn_out; a.ld_attn = ld_attn; a.attn_scale = attn_scale; a.accumulate = accumulate;
    a.N = N; a.H = H; a.heads = heads;
    hipStream_t s = (hipStream_t)stream;
    // One workgroup per image reads each key/value row once for all beams, but measured slower than
    // one workgroup per row on MI355X (25 us vs 20 us at 768 rows, S=49, H=512): kept as an opt-in.
    {
        // wide form: one workgroup of 16 waves per image, every key/value row fetched once for all beams
        static const bool no_wide = getenv("MIC_CROSS_NO_WIDE") != nullptr;
        const int vec = 16 / (int)mic_dtype_size(dtype);
        const size_t lds = sizeof(float) * (4 * (size_t)heads * 64 + 8 * 4 * (size_t)H);
        if (!no_wide && H == 64 * vec && S <= 64 && heads <= 64 && (heads & (heads - 1)) == 0 && rows_per_image >= 2 &&
            N % rows_per_image == 0 && ldq % 4 == 0 && ((uintptr_t)q & 15) == 0 && lds <= 150 * 1024)
            return dtype == MIC_BF16 ? launch_cross_wide<bf16_t>(a, s) : launch_cross_wide<float>(a, s);
    }
    static const bool per_image = getenv("MIC_CROSS_PER_IMAGE") != nullptr;
    if (per_image && rows_per_image > 1 && rows_per_image <= MAXR && N % rows_per_image == 0) {
        // one workgroup per image when its beams' queries, scores and partial outputs fit in 64 KiB of LDS
        const int vec = 16 / (int)mic_dtype_size(dtype), C = H / vec, npg = 256 / C > 0 ? 256 / C : 1, R = rows_per_image;
        const size_t lds = sizeof(float) * ((size_t)R * H + ((R * heads * S + 3) & ~3) + (size_t)npg * R * H);
        if (lds <= 64 * 1024) return dtype == MIC_BF16 ? launch_cross_image<bf16_t>(a, s, lds) : launch_cross_image<float>(a, s, lds);
    }
    return dtype == MIC_BF16 ? launch_attn<bf16_t, false>(a, s) : launch_attn<float, false>(a, s);
}

extern "C" int mic_cross_attn_folded_supported(int dtype, int H, int heads, int S, int rows_per_image) {
    if (dtype != MIC_F32 && dtype != MIC_BF16) return 0;
    const int vec = 16 / (int)mic_dtype_size(dtype);
    return H == 64 * vec && (heads == 1 || heads == 2) && S >= 1 && S <= 64 && rows_per_image >= 1 && rows_per_image <= MIC_MAX_BEAM;
}

extern "C" int mic_cross_fold_bias(const void* kv, const float* q_bias_scaled, float* c, int dtype, int rows, int H, int heads,
                                   void* stream) {
    MIC_ARG_CHECK(kv && q_bias_scaled && c, "mic_cross_fold_bias: null pointer");
    MIC_TRY(check_attn_dims("mic_cross_fold_bias", dtype, rows, H, heads));
    const int grid = (rows * heads + 3) / 4;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(cross_fold_bias_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)kv, q_bias_scaled, c, rows, H, heads);
    else
        MIC_LAUNCH(cross_fold_bias_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)kv, q_bias_scaled, c, rows, H, heads);
    return 0;
}

extern "C" int mic_cross_fold_c(const void* feats, const float* u, const float* c0, float* c, int dtype, int rows, int H, int heads,
                                void* stream) {
    MIC_ARG_CHECK(feats && u && c0 && c, "mic_cross_fold_c: null pointer");
    MIC_TRY(check_attn_dims("mic_cross_fold_c", dtype, rows, H, heads));
    MIC_ARG_CHECK(heads <= MIC_FOLD_MAX_HEADS, "mic_cross_fold_c: %d heads (max %d)", heads, MIC_FOLD_MAX_HEADS);
    MIC_ARG_CHECK(((uintptr_t)feats & 15) == 0, "mic_cross_fold_c: 16-byte alignment");
    const int grid = (rows + 3) / 4;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(cross_fold_c_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feats, u, c0, c, rows, H, heads);
    else
        MIC_LAUNCH(cross_fold_c_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)feats, u, c0, c, rows, H, heads);
    return 0;
}

extern "C" int mic_cross_attn_folded(const mic_ln_ref* x, const void* kf, const void* vf, const float* c, const float* out_bias, int S,
                                     int rows_per_image, float* y, float* stats_out, float* attn_out, int ld_attn, float attn_scale,
                                     int accumulate, int dtype, int N, int H, int heads, void* stream) {
    MIC_ARG_CHECK(x && x->raw && x->stats && x->gamma && x->beta && kf && vf && c && out_bias && y, "mic_cross_attn_folded: null pointer");
    MIC_ARG_CHECK(mic_cross_attn_folded_supported(dtype, H, heads, S, rows_per_image),
                  "mic_cross_attn_folded: unsupported shape (dtype %d, H=%d, heads=%d, S=%d, rows per image %d)", dtype, H, heads, S, rows_per_image);
    MIC_ARG_CHECK(N > 0 && N % rows_per_image == 0, "mic_cross_attn_folded: N=%d is not a multiple of %d rows per image", N, rows_per_image);
    MIC_ARG_CHECK(x->width == H && x->parts == (H + 127) / 128 && x->ld >= H && x->ld % 4 == 0, "mic_cross_attn_folded: LayerNorm view does not match H=%d", H);
    MIC_ARG_CHECK(!attn_out || ld_attn >= S, "mic_cross_attn_folded: ld_attn=%d < S=%d", ld_attn, S);
    MIC_ARG_CHECK(((uintptr_t)x->raw & 15) == 0 && ((uintptr_t)kf & 15) == 0 && ((uintptr_t)vf & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      ((uintptr_t)x->gamma & 15) == 0 && ((uintptr_t)x->beta & 15) == 0,
                  "mic_cross_attn_folded: 16-byte alignment");
    FoldArgs a{};
    a.raw = x->raw; a.stats = x->stats; a.gamma = x->gamma; a.beta = x->beta; a.ld = x->ld; a.parts = x->parts;
    a.kf = kf; a.vf = vf; a.c = c; a.bo = out_bias; a.y = y; a.stats_out = stats_out;
    a.attn_out = attn_out; a.ld_attn = ld_attn; a.attn_scale = attn_scale; a.accumulate = accumulate;
    a.S = S; a.R = rows_per_image; a.N = N; a.H = H;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16) return heads == 1 ? launch_cross_fold<bf16_t, 1>(a, s) : launch_cross_fold<bf16_t, 2>(a, s);
    return heads == 1 ? launch_cross_fold<float, 1>(a, s) : launch_cross_fold<float, 2>(a, s);
}

extern "C" int mic_lstm_cell(const float* gates, float* c, float* h, void* ht0, int ld0, void* ht1, int ld1, int dtype, int N,
                             int Hd, void* stream) {
    MIC_ARG_CHECK(gates && c && h && N > 0 && Hd > 0, "mic_lstm_cell: bad arguments");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_lstm_cell: bad dtype %d", dtype);
    const int total = N * Hd;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(lstm_cell_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, s, gates, c, h, (bf16_t*)ht0, ld0, (bf16_t*)ht1, ld1, N, Hd);
    else
        MIC_LAUNCH(lstm_cell_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, gates, c, h, (float*)ht0, ld0, (float*)ht1, ld1, N, Hd);
    return 0;
}

extern "C" int mic_lstm_cell_ln_supported(int Hd) { return Hd > 0 && Hd % 4 == 0 && Hd <= 1024; }

extern "C" int mic_lstm_cell_ln(const float* gates, float* c, float* h, void* ht0, int ld0, const float* ln_g, const float* ln_b, float* hn,
                                void* hnt, int ld_hnt, int dtype, int N, int Hd, void* stream) {
    MIC_ARG_CHECK(gates && c && h && ln_g && ln_b && (hn || hnt) && N > 0, "mic_lstm_cell_ln: bad arguments");
    MIC_ARG_CHECK(mic_lstm_cell_ln_supported(Hd), "mic_lstm_cell_ln: Hd=%d must be a multiple of 4, at most 1024", Hd);
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_lstm_cell_ln: bad dtype %d", dtype);
    MIC_ARG_CHECK((((uintptr_t)gates | (uintptr_t)c | (uintptr_t)h | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)hn) & 15) == 0,
                  "mic_lstm_cell_ln: fp32 pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((N + 3) / 4), block(256);
#define MIC_CELL_LN(TT, NV) MIC_LAUNCH((lstm_cell_ln_kernel<TT, NV>), grid, block, 0, s, gates, c, h, (TT*)ht0, ld0, ln_g, ln_b, hn, (TT*)hnt, ld_hnt, N, Hd)
    const int nv = (Hd + 255) / 256;
    if (dtype == MIC_BF16) {
        if (nv == 1) MIC_CELL_LN(bf16_t, 1); else if (nv == 2) MIC_CELL_LN(bf16_t, 2); else MIC_CELL_LN(bf16_t, 4);
    } else {
        if (nv == 1) MIC_CELL_LN(float, 1); else if (nv == 2) MIC_CELL_LN(float, 2); else MIC_CELL_LN(float, 4);
    }
#undef MIC_CELL_LN
    return 0;
}

extern "C" int mic_gather_rows(const void* src, void* dst, const int32_t* idx, int rows, int row_bytes, void* stream) {
    MIC_ARG_CHECK(src && dst && idx && rows > 0 && row_bytes > 0 && row_bytes % 4 == 0, "mic_gather_rows: bad arguments");
    MIC_ARG_CHECK(src != dst, "mic_gather_rows: in-place gather is not supported");
    const long long total = (long long)rows * (row_bytes / 4);
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    MIC_LAUNCH(gather_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, (uint32_t*)dst, idx, rows, row_bytes / 4);
    return 0;
}

extern "C" int mic_gather_rows_multi(const void* const* src, void* const* dst, const int32_t* row_bytes, int n_bufs, const int32_t* idx,
                                     int rows, void* stream) {
    MIC_ARG_CHECK(src && dst && row_bytes && idx && rows > 0, "mic_gather_rows_multi: bad arguments");
    MIC_ARG_CHECK(n_bufs >= 1 && n_bufs <= MIC_GATHER_MAX, "mic_gather_rows_multi: %d buffers (1..%d)", n_bufs, MIC_GATHER_MAX);
    GatherMulti g{};
    long long most = 0;
    for (int i = 0; i < n_bufs; ++i) {
        MIC_ARG_CHECK(src[i] && dst[i] && src[i] != dst[i] && row_bytes[i] > 0 && row_bytes[i] % 4 == 0,
                      "mic_gather_rows_multi: buffer %d: null, in-place or row size %d not a multiple of 4", i, row_bytes[i]);
        g.src[i] = (const uint32_t*)src[i]; g.dst[i] = (uint32_t*)dst[i]; g.words[i] = row_bytes[i] / 4;
        const long long total = (long long)rows * g.words[i];
        most = total > most ? total : most;
    }
    long long gx = (most + 255) / 256;
    if (gx > 4096) gx = 4096;
    MIC_LAUNCH(gather_rows_multi_kernel, dim3((unsigned)gx, (unsigned)n_bufs), dim3(256), 0, (hipStream_t)stream, g, idx, rows);
    return 0;
}
