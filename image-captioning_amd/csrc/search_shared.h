// Search pieces shared by the search kernels (search_ops.hip) and the one-launch decoder step (decode_persist.hip), which
// runs the tail of a beam-search step as its last phase: candidate ranking, beam selection, and the tail of a step
// (log-softmax + top-K of an image's rows, selection, the next position's embedding) as a function of a workgroup.
#pragma once
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
// workgroup must call, the barriers are workgroup-wide; `wave` of `nwaves`: this thread's wave among those that call.  The caller has run slot_load / slot_store for the first
// n_live_slots lanes of that wave (no barrier needed in between: the first one below publishes them).  tv / ti: the image's [K][K] top-K values and token ids
// (global or LDS).  Afterwards c_tok[win[r]] is the token appended to surviving slot r, -1 for a finished one.
__device__ __forceinline__ void beam_select_body(const BeamArgs& a, int b, int lane, bool act, SelShared& sh, const float* tv,
                                                 const int32_t* ti, int wave = 0, int nwaves = 1) {
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
    // the survivors' rows move in parallel, one wave each (their loads are chains of dependent round trips)
    for (int r = wave; r < K; r += nwaves) {
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
        }
    }
    if (act && lane == 0 && a.live_images) {
        int any_live = 0;
        for (int r = 0; r < K; ++r) {
            const int tok = sh.c_tok[sh.win[r]];
            any_live |= (tok >= 0 && tok != a.eos_id);
        }
        if (any_live) atomicAdd(a.live_images, 1);
    }
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

template <int K, int G, int NWG> struct TailShared {
    SelShared sh;
    float red[G][NWG];
    KV wbest[G][NWG][K];
    float tv[K * K];
    int32_t ti[K * K];
};

// a logit another workgroup of the same launch wrote (HAND): read past the caches that are not coherent between compute units
template <bool HAND> __device__ __forceinline__ float tail_ld(const float* p) {
    if constexpr (HAND) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// The tail for image `img` by the G * GW threads of a workgroup (thread `tid`).  GW threads work on one row of logits (256 for up to two beams,
// 128 above, so that every row of an image has a group of its own within the 1024 threads of a
// workgroup); a thread holds up to 4096 / GW entries of the row in registers; G rows are worked on at a time.  The result
// does not depend on G.  `wait` is called once, after the loads that do not depend on the logits have been requested and
// before the first logit is read; it returns false to abandon the step (every thread must get the same answer).
template <int K, int GW, int G, bool HAND, class Wait>
__device__ __forceinline__ void beam_tail_body(const TailArgs& a, int img, int tid, TailShared<K, G, GW / 64>& ts, Wait wait) {
    constexpr int R = 4096 / GW, NWG = GW / 64;
    SelShared& sh = ts.sh;
    auto& red = ts.red;
    auto& wbest = ts.wbest;
    float* tv = ts.tv;
    int32_t* ti = ts.ti;
    const int grp = tid / GW, gt = tid % GW, gw = gt >> 6, lane = tid & 63;
    // the beam state of the image's slots does not depend on the logits: requested now, parked in LDS after the top-K passes
    SlotState slot{};
    if (tid < a.b.n_live_slots) slot = slot_load(a.b, img, tid);
    if (!wait()) return;
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
            raw[r] = on && i < a.V ? tail_ld<HAND>(x + i) : -INFINITY;
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
        // each wave extracts its own first K in order (no barrier: the exchange is inside the wave), then the first lanes of
        // the group rank the NWG * K wave candidates: the row's first K are among them, and ranking them gives the order a
        // round-by-round extraction over the whole group would
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const KV wb = wave_best(loc[0]);
            if (lane == 0) wbest[grp][gw][r] = wb;
            if (loc[0].idx == wb.idx && wb.idx != 0x7fffffff) {  // this thread owned the wave's winner: pop
#pragma unroll
                for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
                loc[K - 1] = KV{-INFINITY, 0x7fffffff};
            }
        }
        __syncthreads();
        if (on && gt < NWG * K) {
            const KV mine = wbest[grp][gt / K][gt % K];
            int rank = 0;
#pragma unroll
            for (int o = 0; o < NWG * K; ++o) {
                const KV other = wbest[grp][o / K][o % K];
                rank += before(other, mine) || (other.key == mine.key && other.idx == mine.idx && o < gt);  // the empty entries tie
            }
            if (rank < K) {
                const bool empty = mine.idx == 0x7fffffff;  // fewer than K entries in the row
                tv[row * K + rank] = empty ? -INFINITY : mine.key;
                ti[row * K + rank] = empty ? 0 : mine.idx;
            }
        }
        // (another pass writes wbest only behind the barriers of its own reductions; after the last pass the selection's
        // first barrier publishes tv / ti)
    }
    if (tid < a.b.n_live_slots) slot_store(a.b, sh, tid, slot);
    beam_select_body(a.b, img, lane, tid < 64, sh, tv, ti, tid >> 6, G * NWG);
    if (!a.tok_emb && a.n_gather == 0) return;
    // What follows reads the ranking (win, c_slot, c_tok), final since the selection's last barrier: no barrier here, which
    // would also wait for the row moves the first K waves have in flight (a chain of dependent round trips).  The work is
    // dealt from the last thread down, so that it falls to the waves that move no row first.
    const int rtid = GW * G - 1 - tid;
    // survivors' state rows: slot r of the next step continues slot c_slot[win[r]] of this one
    for (int g = 0; g < a.n_gather; ++g) {
        const int words = a.g_words[g];
        for (int i = rtid; i < K * words; i += GW * G) {
            const int r = i / words, w = i - r * words;
            const long long src_row = (long long)img * K + sh.c_slot[sh.win[r]], dst_row = (long long)img * K + r;
            a.g_dst[g][dst_row * a.g_stride[g] + w] = a.g_src[g][src_row * a.g_stride[g] + w];
        }
    }
    if (!a.tok_emb) return;
    const int hv = a.H >> 2, pos = a.b.t + 1;
    for (int i = rtid; i < K * hv; i += GW * G) {
        const int r = i / hv, c = (i - r * hv) << 2;
        int id = sh.c_tok[sh.win[r]];  // -1: a finished beam
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

// Checks a tail descriptor and turns it into the kernel's arguments.
inline int tail_args_from_desc(const mic_beam_tail_desc* d, TailArgs& a) {
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
    a = TailArgs{};
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
    return 0;
}

}  // namespace
