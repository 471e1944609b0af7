// Search kernels: log-softmax + top-K over the vocabulary, inverse-CDF sampling, beam selection /
// reorder, best-beam extraction, greedy bookkeeping.  Pure bandwidth / latency work: one pass of
// wavefront reductions per row of logits, and integer + fp64 bookkeeping per image.
#include "mic_common.h"

namespace {

constexpr int KMAX = MIC_MAX_BEAM;

struct KV {
    float key;
    int idx;
};
// a ranks before b: larger key first, lower index on ties
__device__ __forceinline__ bool before(const KV& a, const KV& b) { return a.key > b.key || (a.key == b.key && a.idx < b.idx); }

// The wave's first-ranked entry, returned to every lane: quad permutes and row mirrors on the data-parallel-primitive path
// within each row of 16 lanes (as wave_max), then the four row winners are read into scalar registers and ranked there.
__device__ __forceinline__ KV wave_best(KV v) {
#define MIC_BEST_STEP(ctrl)                                                                                   \
    {                                                                                                         \
        KV t;                                                                                                 \
        t.key = MIC_DPP_F32(v.key, ctrl);                                                                     \
        t.idx = __builtin_amdgcn_update_dpp(0, v.idx, ctrl, 0xf, 0xf, false);                                 \
        if (before(t, v)) v = t;                                                                              \
    }
    MIC_BEST_STEP(0xB1)   // quad_perm [1,0,3,2]
    MIC_BEST_STEP(0x4E)   // quad_perm [2,3,0,1]
    MIC_BEST_STEP(0x141)  // row_half_mirror
    MIC_BEST_STEP(0x140)  // row_mirror: every lane holds the best of its row of 16
#undef MIC_BEST_STEP
    KV r = KV{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.key), 0)), __builtin_amdgcn_readlane(v.idx, 0)};
#pragma unroll
    for (int l = 16; l < 64; l += 16) {
        const KV t{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.key), l)), __builtin_amdgcn_readlane(v.idx, l)};
        if (before(t, r)) r = t;
    }
    return r;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one 256-thread workgroup per row of logits
template <int K>
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                              int rank_raw, float* __restrict__ out_val,
                                                              int32_t* __restrict__ out_idx, float* __restrict__ lse) {
    __shared__ float red[4];
    __shared__ KV wbest[4];
    __shared__ int winner;
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;

    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i] * inv_temp);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(x[i] * inv_temp - m);
    s = block_sum(s, red);
    const float logs = logf(s);
    if (lse && tid == 0) lse[blockIdx.x] = m + logs;

    // thread-local sorted top-K of its strided elements (ascending index order => strict '>' keeps
    // the lower index ahead on ties)
    KV loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) loc[k] = KV{-INFINITY, 0x7fffffff};
    for (int i = tid; i < V; i += 256) {
        const float xs = x[i] * inv_temp;
        KV c{rank_raw ? x[i] : (xs - m) - logs, i};
        if (before(c, loc[K - 1])) {
            loc[K - 1] = c;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
                if (before(loc[k], loc[k - 1])) {
                    const KV t = loc[k - 1];
                    loc[k - 1] = loc[k];
                    loc[k] = t;
                }
            }
        }
    }
    // K rounds of block-wide selection over the heads of the per-thread lists
    for (int r = 0; r < K; ++r) {
        const KV b = wave_best(loc[0]);
        if ((tid & 63) == 0) wbest[tid >> 6] = b;
        __syncthreads();
        KV best = wbest[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (before(wbest[w], best)) best = wbest[w];
        if (loc[0].idx == best.idx && best.idx != 0x7fffffff) {
            // this thread owned the winner: emit and pop
            if (tid >= 0) {
                const float xs = x[best.idx] * inv_temp;
                out_val[(long long)blockIdx.x * K + r] = (xs - m) - logs;
                out_idx[(long long)blockIdx.x * K + r] = best.idx;
            }
#pragma unroll
            for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
            loc[K - 1] = KV{-INFINITY, 0x7fffffff};
        } else if (best.idx == 0x7fffffff && tid == 0) {
            out_val[(long long)blockIdx.x * K + r] = -INFINITY;  // fewer than K entries in the row
            out_idx[(long long)blockIdx.x * K + r] = 0;
        }
        __syncthreads();
    }
    (void)winner;
}

// Same result for rows of at most 256 * R entries, read from memory ONCE: every load of the row is issued
// up front (one memory round trip instead of three passes of dependent ones) and the maximum, the sum
// and the selection all work on registers.
template <int K, int R>
__global__ __launch_bounds__(256) void logsoftmax_topk_reg_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                                  int rank_raw, float* __restrict__ out_val,
                                                                  int32_t* __restrict__ out_idx, float* __restrict__ lse) {
    __shared__ float red[4];
    __shared__ KV wbest[4];
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;
    float raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + 256 * r;
        raw[r] = i < V ? x[i] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) m = fmaxf(m, raw[r] * inv_temp);
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tid + 256 * r < V) s += expf(raw[r] * inv_temp - m);
    s = block_sum(s, red);
    const float logs = logf(s);
    if (lse && tid == 0) lse[blockIdx.x] = m + logs;

    KV loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) loc[k] = KV{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + 256 * r;
        if (i < V) {
            KV c{rank_raw ? raw[r] : (raw[r] * inv_temp - m) - logs, i};
            if (before(c, loc[K - 1])) {
                loc[K - 1] = c;
#pragma unroll
                for (int k = K - 1; k > 0; --k) {
                    if (before(loc[k], loc[k - 1])) {
                        const KV t = loc[k - 1];
                        loc[k - 1] = loc[k];
                        loc[k] = t;
                    }
                }
            }
        }
    }
    for (int r = 0; r < K; ++r) {
        const KV b = wave_best(loc[0]);
        if ((tid & 63) == 0) wbest[tid >> 6] = b;
        __syncthreads();
        KV best = wbest[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (before(wbest[w], best)) best = wbest[w];
        if (loc[0].idx == best.idx && best.idx != 0x7fffffff) {
            // this thread owned the winner: emit (always as a log-probability) and pop
            const float xs = x[best.idx] * inv_temp;
            out_val[(long long)blockIdx.x * K + r] = (xs - m) - logs;
            out_idx[(long long)blockIdx.x * K + r] = best.idx;
#pragma unroll
            for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
            loc[K - 1] = KV{-INFINITY, 0x7fffffff};
        } else if (best.idx == 0x7fffffff && tid == 0) {
            out_val[(long long)blockIdx.x * K + r] = -INFINITY;  // fewer than K entries in the row
            out_idx[(long long)blockIdx.x * K + r] = 0;
        }
        __syncthreads();
    }
}

// one 256-thread workgroup per row; thread i owns the contiguous slice [i*chunk, (i+1)*chunk)
__global__ __launch_bounds__(256) void sample_multinomial_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                                 const float* __restrict__ u, float* __restrict__ out_val,
                                                                 int32_t* __restrict__ out_idx) {
    __shared__ float red[4];
    __shared__ double scan[256];
    __shared__ int first;
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i] * inv_temp);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(x[i] * inv_temp - m);
    s = block_sum(s, red);
    const float logs = logf(s);
    const int chunk = (V + 255) / 256;
    const int lo = tid * chunk, hi = min(V, lo + chunk);
    double part = 0.0;
    for (int i = lo; i < hi; ++i) part += (double)expf((x[i] * inv_temp - m) - logs);
    scan[tid] = part;
    if (tid == 0) first = V - 1;
    __syncthreads();
    if (tid == 0) {  // 256-entry exclusive scan, serial: negligible next to the row reads
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double t = scan[i];
            scan[i] = run;
            run += t;
        }
    }
    __syncthreads();
    const double uu = (double)u[blockIdx.x];
    double run = scan[tid];
    for (int i = lo; i < hi; ++i) {
        run += (double)expf((x[i] * inv_temp - m) - logs);
        if (run > uu) {
            atomicMin(&first, i);
            break;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int i = first;
        out_idx[blockIdx.x] = i;
        out_val[blockIdx.x] = (x[i] * inv_temp - m) - logs;
    }
}

// ---------------------------------------------------------------------------------------------
// beam selection: one wave per image
struct BeamArgs {
    mic_beam_state cur, nxt;
    const float* top_val;
    const int32_t* top_idx;
    int32_t* parent;
    int32_t* live_images;
    const double* sqrt_tab;
    int B, K, n_live_slots, t, t_cap, eos_id, in_place;
};

// A NaN key compares false against everything, which would give several candidates rank 0 and leave
// other ranks without a winner; it sorts as -infinity instead (ties then fall back to candidate order).
__device__ __forceinline__ double sane_key(double k) { return k == k ? k : -INFINITY; }

struct SelShared {
    double c_key[KMAX * KMAX + KMAX], c_score[KMAX * KMAX + KMAX];
    int c_slot[KMAX * KMAX + KMAX], c_tok[KMAX * KMAX + KMAX];
    int s_fin[KMAX], s_off[KMAX], win[KMAX];
    // per live slot, read from the beam state before anything else: length, score, sqrt_tab[len] and sqrt_tab[len + 1]
    int s_len[KMAX];
    double s_score[KMAX], s_sq0[KMAX], s_sq1[KMAX];
};

// What the selection needs of slot `lane` of image b, requested together (two dependent round trips: the length, then the
// last token and the square roots); the caller parks the values in LDS with slot_store once they are due.
struct SlotState {
    int len, last;
    double score, sq0, sq1;
};
__device__ __forceinline__ SlotState slot_load(const BeamArgs& a, int b, int lane) {
    const int g = b * a.K + lane;
    SlotState s;
    s.len = a.cur.len[g];
    s.score = a.cur.score[g];
    s.last = a.cur.seq[(long long)g * a.t_cap + s.len - 1];
    s.sq0 = a.sqrt_tab[s.len];
    s.sq1 = a.sqrt_tab[s.len + 1];
    return s;
}
__device__ __forceinline__ void slot_store(const BeamArgs& a, SelShared& sh, int lane, const SlotState& s) {
    sh.s_fin[lane] = s.last == a.eos_id;
    sh.s_len[lane] = s.len;
    sh.s_score[lane] = s.score;
    sh.s_sq0[lane] = s.sq0;
    sh.s_sq1[lane] = s.sq1;
}

// Selection for image b by one wave (`act`: this thread belongs to it, `lane` its lane); every thread of the
// workgroup must call, the barriers are workgroup-wide.  The caller has run slot_load / slot_store for the first
// n_live_slots lanes of that wave (no barrier needed in between: the first one below publishes them).  tv / ti: the image's [K][K] top-K values and token ids
// (global or LDS).  new_tok (LDS, [K], may be null): token appended to each surviving slot, -1 for a finished one.
__device__ __forceinline__ void beam_select_body(const BeamArgs& a, int b, int lane, bool act, SelShared& sh, const float* tv,
                                                 const int32_t* ti, int* new_tok) {
    const int K = a.K;
    const int base = b * K;
    const int ns = a.n_live_slots;
    __syncthreads();
    int n_live = 0, n_fin = 0;
    for (int i = 0; i < ns; ++i) (sh.s_fin[i] ? n_fin : n_live)++;
    if (act && lane == 0) {
        // position of each slot's first candidate in the reference's candidate list
        int off = 0, lv = 0, fn = 0;
        for (int i = 0; i < ns; ++i) {
            if (a.in_place) {
                sh.s_off[i] = off;
                off += sh.s_fin[i] ? 1 : K;
            } else {
                sh.s_off[i] = sh.s_fin[i] ? n_live * K + fn : lv * K;
                (sh.s_fin[i] ? fn : lv)++;
            }
        }
    }
    if (act && lane < K) sh.win[lane] = lane;  // every rank has a winner whatever the keys are
    __syncthreads();
    const int ncand = n_live * K + n_fin;
    // fill candidates: slot i contributes K entries (live) or itself (finished)
    if (act) {
        for (int w = lane; w < ns * K; w += 64) {
            const int i = w / K, j = w - i * K;
            if (sh.s_fin[i]) {
                if (j == 0) {
                    const int c = sh.s_off[i];
                    const double sc = sh.s_score[i];
                    sh.c_score[c] = sc;
                    sh.c_key[c] = sane_key(sc / sh.s_sq0[i]);
                    sh.c_slot[c] = i;
                    sh.c_tok[c] = -1;
                }
            } else {
                const int c = sh.s_off[i] + j;
                const double sc = sh.s_score[i] + (double)tv[i * K + j];
                sh.c_score[c] = sc;
                sh.c_key[c] = sane_key(sc / sh.s_sq1[i]);
                sh.c_slot[c] = i;
                sh.c_tok[c] = ti[i * K + j];
            }
        }
    }
    __syncthreads();
    // stable descending rank = number of candidates that sort strictly ahead
    if (act) {
        for (int c = lane; c < ncand; c += 64) {
            const double k = sh.c_key[c];
            int rank = 0;
            for (int o = 0; o < ncand; ++o) {
                const double ko = sh.c_key[o];
                rank += (ko > k) || (ko == k && o < c);
            }
            if (rank < K) sh.win[rank] = c;
        }
    }
    __syncthreads();
    if (!act) return;
    int any_live = 0;
    for (int r = 0; r < K; ++r) {
        const int c = sh.win[r], i = sh.c_slot[c], tok = sh.c_tok[c];
        const int gs = base + i, gd = base + r;
        const int len = sh.s_len[i];
        const int32_t* sseq = a.cur.seq + (long long)gs * a.t_cap;
        int32_t* dseq = a.nxt.seq + (long long)gd * a.t_cap;
        for (int p = lane; p < len; p += 64) dseq[p] = sseq[p];
        if (a.cur.anc) {
            const int32_t* sa = a.cur.anc + (long long)gs * a.t_cap;
            int32_t* da = a.nxt.anc + (long long)gd * a.t_cap;
            const int na = tok >= 0 ? a.t : len - 1;  // positions this lineage has been run for
            for (int p = lane; p < na; p += 64) da[p] = sa[p];
            if (tok >= 0 && lane == 0) da[a.t] = gs;
        }
        if (lane == 0) {
            a.nxt.score[gd] = sh.c_score[c];
            a.nxt.len[gd] = tok >= 0 ? len + 1 : len;
            if (tok >= 0) dseq[len] = tok;
            a.parent[gd] = gs;
            if (new_tok) new_tok[r] = tok;
        }
        any_live |= (tok >= 0 && tok != a.eos_id);
    }
    if (lane == 0 && any_live && a.live_images) atomicAdd(a.live_images, 1);
}

__global__ __launch_bounds__(64) void beam_select_kernel(const BeamArgs a) {
    __shared__ SelShared sh;
    const long long base = (long long)blockIdx.x * a.K;
    if ((int)threadIdx.x < a.n_live_slots) slot_store(a, sh, threadIdx.x, slot_load(a, blockIdx.x, threadIdx.x));
    beam_select_body(a, blockIdx.x, threadIdx.x, true, sh, a.top_val + base * a.K, a.top_idx + base * a.K, nullptr);
}

// The tail of a beam-search step in one launch per image: log-softmax + top-K of the image's K rows of logits (one group
// of 256 threads per row, the row read from memory once into registers), beam selection by the first wave, and the
// embedding of the chosen tokens for the next position.  Unfused these are three launches, each a chain of dependent
// memory round trips behind a kernel boundary on a path that is latency bound.
struct TailArgs {
    BeamArgs b;
    const float* logits;
    int ld, V;
    const float *tok_emb, *pos_emb;  // embedding for position b.t + 1 is skipped when tok_emb is null
    float* x;
    void* xt;
    int H, vocab, xt_bf16, ld_xt;
    // per-beam state that travels with the survivors: dst[g][row r of the image] = src[g][parent of r], n_gather buffers
    int n_gather;
    const uint32_t* g_src[MIC_GATHER_MAX];
    uint32_t* g_dst[MIC_GATHER_MAX];
    int g_words[MIC_GATHER_MAX], g_stride[MIC_GATHER_MAX];  // 4-byte words per row to move / between rows
};

// GW threads work on one row of logits (256, or 128 for more than four beams so that every row of the image has a
// group of its own within the 1024 threads of a workgroup); a thread holds up to 4096 / GW entries of the row in registers.
template <int K, int GW>
__global__ __launch_bounds__(GW * (K < 1024 / GW ? K : 1024 / GW)) void beam_tail_kernel(const TailArgs a) {
    constexpr int G = K < 1024 / GW ? K : 1024 / GW, R = 4096 / GW, NWG = GW / 64;
    __shared__ SelShared sh;
    __shared__ float red[G][NWG];
    __shared__ KV wbest[G][NWG];
    __shared__ float tv[K * K];
    __shared__ int32_t ti[K * K];
    __shared__ int new_tok[K];
    const int tid = threadIdx.x, grp = tid / GW, gt = tid % GW, gw = gt >> 6, lane = tid & 63;
    const int img = blockIdx.x;
    // the beam state of the image's slots does not depend on the logits: requested now, parked in LDS after the top-K passes
    SlotState slot{};
    if (tid < a.b.n_live_slots) slot = slot_load(a.b, img, tid);
    auto group_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[grp][gw] = v;
        __syncthreads();
        float m = red[grp][0];
#pragma unroll
        for (int w = 1; w < NWG; ++w) m = fmaxf(m, red[grp][w]);
        return m;
    };
    auto group_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[grp][gw] = v;
        __syncthreads();
        if constexpr (NWG == 4) return (red[grp][0] + red[grp][1]) + (red[grp][2] + red[grp][3]);
        else return red[grp][0] + red[grp][1];
    };
    for (int row0 = 0; row0 < K; row0 += G) {
        const int row = row0 + grp;
        const bool on = row < K;  // every group runs the same barriers; one without a row works on -infinity
        const float* x = a.logits + ((long long)img * K + (on ? row : 0)) * a.ld;
        float raw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = gt + GW * r;
            raw[r] = on && i < a.V ? x[i] : -INFINITY;
        }
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r) m = fmaxf(m, raw[r]);
        m = group_max(m);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (on && gt + GW * r < a.V) s += expf(raw[r] - m);
        s = group_sum(s);
        const float logs = logf(s);
        KV loc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) loc[k] = KV{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = gt + GW * r;
            if (on && i < a.V) {
                KV c{(raw[r] - m) - logs, i};
                if (before(c, loc[K - 1])) {
                    loc[K - 1] = c;
#pragma unroll
                    for (int k = K - 1; k > 0; --k) {
                        if (before(loc[k], loc[k - 1])) {
                            const KV t = loc[k - 1];
                            loc[k - 1] = loc[k];
                            loc[k] = t;
                        }
                    }
                }
            }
        }
        for (int r = 0; r < K; ++r) {
            const KV wb = wave_best(loc[0]);
            if (lane == 0) wbest[grp][gw] = wb;
            __syncthreads();
            KV best = wbest[grp][0];
#pragma unroll
            for (int w = 1; w < NWG; ++w)
                if (before(wbest[grp][w], best)) best = wbest[grp][w];
            if (on) {
                if (loc[0].idx == best.idx && best.idx != 0x7fffffff) {  // this thread owned the winner: emit and pop
                    tv[row * K + r] = best.key;
                    ti[row * K + r] = best.idx;
#pragma unroll
                    for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
                    loc[K - 1] = KV{-INFINITY, 0x7fffffff};
                } else if (best.idx == 0x7fffffff && gt == 0) {  // fewer than K entries in the row
                    tv[row * K + r] = -INFINITY;
                    ti[row * K + r] = 0;
                }
            }
            __syncthreads();
        }
    }
    if (tid < a.b.n_live_slots) slot_store(a.b, sh, tid, slot);
    beam_select_body(a.b, img, lane, tid < 64, sh, tv, ti, new_tok);
    if (!a.tok_emb && a.n_gather == 0) return;
    __syncthreads();
    // survivors' state rows: slot r of the next step continues slot c_slot[win[r]] of this one
    for (int g = 0; g < a.n_gather; ++g) {
        const int words = a.g_words[g];
        for (int i = tid; i < K * words; i += GW * G) {
            const int r = i / words, w = i - r * words;
            const long long src_row = (long long)img * K + sh.c_slot[sh.win[r]], dst_row = (long long)img * K + r;
            a.g_dst[g][dst_row * a.g_stride[g] + w] = a.g_src[g][src_row * a.g_stride[g] + w];
        }
    }
    if (!a.tok_emb) return;
    const int hv = a.H >> 2, pos = a.b.t + 1;
    for (int i = tid; i < K * hv; i += GW * G) {
        const int r = i / hv, c = (i - r * hv) << 2;
        int id = new_tok[r];
        id = id < 0 ? a.b.eos_id : id;  // a finished beam's row is never looked at again: any valid token serves
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        float4 v = *reinterpret_cast<const float4*>(a.tok_emb + (long long)id * a.H + c);
        if (a.pos_emb) {
            const float4 p = *reinterpret_cast<const float4*>(a.pos_emb + (long long)pos * a.H + c);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const long long n = (long long)img * K + r;
        if (a.x) *reinterpret_cast<float4*>(a.x + n * a.H + c) = v;
        if (a.xt) {
            if (a.xt_bf16) {
                bf16_t* o = (bf16_t*)a.xt + n * a.ld_xt + c;
                o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
            } else {
                *reinterpret_cast<float4*>((float*)a.xt + n * a.ld_xt + c) = v;
            }
        }
    }
}

__global__ __launch_bounds__(64) void beam_finalize_kernel(const mic_beam_state st, const double* __restrict__ sqrt_tab, int B, int K,
                                                           int t_cap, int32_t* __restrict__ tokens, int32_t* __restrict__ lengths,
                                                           float* __restrict__ score, const float* __restrict__ hist, int S,
                                                           float* __restrict__ attn_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int best = 0;
    double bk = st.score[b * K] / sqrt_tab[st.len[b * K]];
    for (int r = 1; r < K; ++r) {
        const double k = st.score[b * K + r] / sqrt_tab[st.len[b * K + r]];
        if (k > bk) { bk = k; best = r; }
    }
    const int g = b * K + best, len = st.len[g];
    for (int p = lane; p < t_cap; p += 64) tokens[(long long)b * t_cap + p] = p < len ? st.seq[(long long)g * t_cap + p] : -1;
    if (lane == 0) {
        lengths[b] = len;
        score[b] = (float)st.score[g];
    }
    if (hist && attn_out && st.anc) {
        for (int s = lane; s < S; s += 64) {
            float run = 0.f;
            for (int p = 0; p < len - 1; ++p) {
                const int src = st.anc[(long long)g * t_cap + p];
                run += hist[((long long)src * t_cap + p) * S + s];
                attn_out[((long long)b * t_cap + p) * S + s] = run / (float)(p + 1);
            }
        }
    }
}

__global__ void sample_update_kernel(const float* __restrict__ val, const int32_t* __restrict__ idx, int32_t* __restrict__ tokens,
                                     int t_cap, int pos, uint8_t* __restrict__ finished, float* __restrict__ total,
                                     int32_t* live_rows, int N, int pad_id, int eos_id) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool fin = finished[n] != 0;
    if (total && !fin) total[n] += val[n];
    const int tok = fin ? pad_id : idx[n];
    tokens[(long long)n * t_cap + pos] = tok;
    const bool nf = fin || tok == eos_id;
    finished[n] = nf;
    if (!nf && live_rows) atomicAdd(live_rows, 1);
}

}  // namespace

extern "C" int mic_logsoftmax_topk(const float* logits, int ld, int N, int V, int K, float inv_temp, int rank_raw, float* out_val,
                                   int32_t* out_idx, float* lse, void* stream) {
    MIC_ARG_CHECK(logits && out_val && out_idx, "mic_logsoftmax_topk: null pointer");
    MIC_ARG_CHECK(N > 0 && V > 0 && ld >= V, "mic_logsoftmax_topk: bad sizes N=%d V=%d ld=%d", N, V, ld);
    MIC_ARG_CHECK(K >= 1 && K <= KMAX, "mic_logsoftmax_topk: K=%d outside [1,%d]", K, KMAX);
    hipStream_t s = (hipStream_t)stream;
#define MIC_TOPK_CASE(KK)                                                                                                  \
    case KK:                                                                                                               \
        if (V <= 256 * 16)                                                                                                 \
            MIC_LAUNCH((logsoftmax_topk_reg_kernel<KK, 16>), dim3(N), dim3(256), 0, s, logits, ld, V, inv_temp, rank_raw,  \
                       out_val, out_idx, lse);                                                                             \
        else                                                                                                               \
            MIC_LAUNCH(logsoftmax_topk_kernel<KK>, dim3(N), dim3(256), 0, s, logits, ld, V, inv_temp, rank_raw, out_val,   \
                       out_idx, lse);                                                                                      \
        break;
    switch (K) {
        MIC_TOPK_CASE(1) MIC_TOPK_CASE(2) MIC_TOPK_CASE(3) MIC_TOPK_CASE(4)
        MIC_TOPK_CASE(5) MIC_TOPK_CASE(6) MIC_TOPK_CASE(7) MIC_TOPK_CASE(8)
    }
#undef MIC_TOPK_CASE
    return 0;
}

extern "C" int mic_sample_multinomial(const float* logits, int ld, int N, int V, float inv_temp, const float* u, float* out_val,
                                      int32_t* out_idx, void* stream) {
    MIC_ARG_CHECK(logits && u && out_val && out_idx, "mic_sample_multinomial: null pointer");
    MIC_ARG_CHECK(N > 0 && V > 0 && ld >= V, "mic_sample_multinomial: bad sizes N=%d V=%d ld=%d", N, V, ld);
    MIC_LAUNCH(sample_multinomial_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, ld, V, inv_temp, u, out_val, out_idx);
    return 0;
}

extern "C" int mic_beam_select(const mic_beam_state* cur, const mic_beam_state* nxt, const float* top_val, const int32_t* top_idx,
                               int32_t* parent, int32_t* live_images, const double* sqrt_tab, int B, int K, int n_live_slots, int t,
                               int t_cap, int eos_id, int finished_in_place, void* stream) {
    MIC_ARG_CHECK(cur && nxt && top_val && top_idx && parent && sqrt_tab, "mic_beam_select: null pointer");
    MIC_ARG_CHECK(cur->score && cur->len && cur->seq && nxt->score && nxt->len && nxt->seq, "mic_beam_select: null state");
    MIC_ARG_CHECK((cur->anc == nullptr) == (nxt->anc == nullptr), "mic_beam_select: anc must be set in both states or neither");
    MIC_ARG_CHECK(B > 0 && K >= 1 && K <= KMAX, "mic_beam_select: B=%d K=%d (K <= %d)", B, K, KMAX);
    MIC_ARG_CHECK(n_live_slots >= 1 && n_live_slots <= K, "mic_beam_select: n_live_slots=%d", n_live_slots);
    MIC_ARG_CHECK(t >= 0 && t + 2 <= t_cap, "mic_beam_select: step %d does not fit t_cap=%d", t, t_cap);
    BeamArgs a;
    a.cur = *cur; a.nxt = *nxt; a.top_val = top_val; a.top_idx = top_idx; a.parent = parent; a.live_images = live_images;
    a.sqrt_tab = sqrt_tab; a.B = B; a.K = K; a.n_live_slots = n_live_slots; a.t = t; a.t_cap = t_cap; a.eos_id = eos_id;
    a.in_place = finished_in_place;
    MIC_LAUNCH(beam_select_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    return 0;
}

extern "C" int mic_beam_tail_supported(int K, int V) { return K >= 1 && K <= KMAX && V >= 1 && V <= 256 * 16; }

extern "C" int mic_beam_tail_ex(const mic_beam_tail_desc* d, void* stream) {
    MIC_ARG_CHECK(d && d->logits && d->cur && d->nxt && d->parent && d->sqrt_tab, "mic_beam_tail: null pointer");
    const mic_beam_state *cur = d->cur, *nxt = d->nxt;
    const int B = d->B, K = d->K, V = d->V;
    MIC_ARG_CHECK(cur->score && cur->len && cur->seq && nxt->score && nxt->len && nxt->seq, "mic_beam_tail: null state");
    MIC_ARG_CHECK((cur->anc == nullptr) == (nxt->anc == nullptr), "mic_beam_tail: anc must be set in both states or neither");
    MIC_ARG_CHECK(B > 0 && mic_beam_tail_supported(K, V) && d->ld >= V, "mic_beam_tail: B=%d K=%d V=%d ld=%d (K <= %d, V <= 4096)", B, K, V, d->ld, KMAX);
    MIC_ARG_CHECK(d->n_live_slots >= 1 && d->n_live_slots <= K, "mic_beam_tail: n_live_slots=%d", d->n_live_slots);
    MIC_ARG_CHECK(d->t >= 0 && d->t + 2 <= d->t_cap, "mic_beam_tail: step %d does not fit t_cap=%d", d->t, d->t_cap);
    MIC_ARG_CHECK(!d->tok_emb || ((d->x || d->xt) && d->H > 0 && d->H % 4 == 0 && d->vocab > 0 && (d->dtype == MIC_F32 || d->dtype == MIC_BF16) &&
                                  (!d->xt || (d->ld_xt >= d->H && d->ld_xt % 4 == 0))),
                  "mic_beam_tail: the embedding of the next position needs an output, H %% 4 == 0 and ld_xt >= H");
    MIC_ARG_CHECK(d->n_gather >= 0 && d->n_gather <= MIC_GATHER_MAX, "mic_beam_tail: %d gathered buffers (max %d)", d->n_gather, MIC_GATHER_MAX);
    TailArgs a{};
    a.b.cur = *cur; a.b.nxt = *nxt; a.b.parent = d->parent; a.b.live_images = d->live_images; a.b.sqrt_tab = d->sqrt_tab;
    a.b.B = B; a.b.K = K; a.b.n_live_slots = d->n_live_slots; a.b.t = d->t; a.b.t_cap = d->t_cap; a.b.eos_id = d->eos_id;
    a.b.in_place = d->finished_in_place;
    a.logits = d->logits; a.ld = d->ld; a.V = V; a.tok_emb = d->tok_emb; a.pos_emb = d->pos_emb; a.x = d->x; a.xt = d->xt; a.H = d->H;
    a.vocab = d->vocab; a.ld_xt = d->ld_xt;
    a.xt_bf16 = d->dtype == MIC_BF16;
    a.n_gather = d->n_gather;
    for (int g = 0; g < d->n_gather; ++g) {
        MIC_ARG_CHECK(d->gather_src[g] && d->gather_dst[g] && d->gather_src[g] != d->gather_dst[g] && d->gather_row_bytes[g] > 0 &&
                          d->gather_row_bytes[g] % 4 == 0 && d->gather_stride_bytes[g] % 4 == 0 && d->gather_stride_bytes[g] >= d->gather_row_bytes[g] &&
                          (((uintptr_t)d->gather_src[g] | (uintptr_t)d->gather_dst[g]) & 3) == 0,
                      "mic_beam_tail: gathered buffer %d: null, in place, or sizes not multiples of 4 bytes", g);
        a.g_src[g] = (const uint32_t*)d->gather_src[g]; a.g_dst[g] = (uint32_t*)d->gather_dst[g];
        a.g_words[g] = d->gather_row_bytes[g] / 4; a.g_stride[g] = d->gather_stride_bytes[g] / 4;
    }
    hipStream_t s = (hipStream_t)stream;
#define MIC_TAIL_CASE(KK) \
    case KK: MIC_LAUNCH((beam_tail_kernel<KK, (KK <= 4 ? 256 : 128)>), dim3(B), dim3(KK <= 4 ? 256 * KK : 128 * KK), 0, s, a); break;
    switch (K) {
        MIC_TAIL_CASE(1) MIC_TAIL_CASE(2) MIC_TAIL_CASE(3) MIC_TAIL_CASE(4)
        MIC_TAIL_CASE(5) MIC_TAIL_CASE(6) MIC_TAIL_CASE(7) MIC_TAIL_CASE(8)
    }
#undef MIC_TAIL_CASE
    return 0;
}

extern "C" int mic_beam_tail(const float* logits, int ld, int V, const mic_beam_state* cur, const mic_beam_state* nxt, int32_t* parent,
                             int32_t* live_images, const double* sqrt_tab, int B, int K, int n_live_slots, int t, int t_cap, int eos_id,
                             const float* tok_emb, const float* pos_emb, float* x, void* xt, int dtype, int H, int vocab, void* stream) {
    MIC_ARG_CHECK(!tok_emb || pos_emb, "mic_beam_tail: the embedding of the next position needs pos_emb");
    mic_beam_tail_desc d{};
    d.logits = logits; d.ld = ld; d.V = V; d.cur = cur; d.nxt = nxt; d.parent = parent; d.live_images = live_images; d.sqrt_tab = sqrt_tab;
    d.B = B; d.K = K; d.n_live_slots = n_live_slots; d.t = t; d.t_cap = t_cap; d.eos_id = eos_id;
    d.tok_emb = tok_emb; d.pos_emb = pos_emb; d.x = x; d.xt = xt; d.ld_xt = H; d.dtype = dtype; d.H = H; d.vocab = vocab;
    return mic_beam_tail_ex(&d, stream);
}

namespace {
__global__ void beam_init_kernel(const mic_beam_state a, const mic_beam_state b, int32_t* __restrict__ live, int N, int t_cap, int sos) {
    const long long total = (long long)N * t_cap;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        a.seq[i] = sos; b.seq[i] = sos;
        if (a.anc) a.anc[i] = 0;
        if (b.anc) b.anc[i] = 0;
        if (i < N) { a.score[i] = 0.0; b.score[i] = 0.0; a.len[i] = 1; b.len[i] = 1; }
        if (live && i < t_cap) live[i] = 0;
    }
}
}  // namespace

extern "C" int mic_beam_init(const mic_beam_state st[2], int32_t* live_images, int N, int t_cap, int sos_id, void* stream) {
    MIC_ARG_CHECK(st && N > 0 && t_cap > 0, "mic_beam_init: bad arguments");
    for (int i = 0; i < 2; ++i) MIC_ARG_CHECK(st[i].score && st[i].len && st[i].seq, "mic_beam_init: null state");
    long long g = ((long long)N * t_cap + 255) / 256;
    if (g > 2048) g = 2048;
    MIC_LAUNCH(beam_init_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, st[0], st[1], live_images, N, t_cap, sos_id);
    return 0;
}

extern "C" int mic_beam_finalize(const mic_beam_state* st, const double* sqrt_tab, int B, int K, int t_cap, int32_t* tokens,
                                 int32_t* lengths, float* score, const float* attn_hist, int S, float* attn_out, void* stream) {
    MIC_ARG_CHECK(st && st->score && st->len && st->seq && sqrt_tab && tokens && lengths && score, "mic_beam_finalize: null pointer");
    MIC_ARG_CHECK(B > 0 && K >= 1 && K <= KMAX && t_cap > 0, "mic_beam_finalize: bad sizes");
    MIC_ARG_CHECK(!attn_out || (attn_hist && st->anc && S > 0), "mic_beam_finalize: attention output needs hist, anc and S");
    MIC_LAUNCH(beam_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *st, sqrt_tab, B, K, t_cap, tokens, lengths, score,
                       attn_hist, S, attn_out);
    return 0;
}

extern "C" int mic_sample_update(const float* val, const int32_t* idx, int32_t* tokens, int t_cap, int pos, uint8_t* finished,
                                 float* total, int32_t* live_rows, int N, int pad_id, int eos_id, void* stream) {
    MIC_ARG_CHECK(idx && tokens && finished && N > 0, "mic_sample_update: bad arguments");
    MIC_ARG_CHECK(!total || val, "mic_sample_update: total needs val");
    MIC_ARG_CHECK(pos >= 0 && pos < t_cap, "mic_sample_update: pos=%d outside t_cap=%d", pos, t_cap);
    MIC_LAUNCH(sample_update_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, val, idx, tokens, t_cap, pos,
                       finished, total, live_rows, N, pad_id, eos_id);
    return 0;
}
